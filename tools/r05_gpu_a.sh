#!/bin/bash
# round 5, GPU call A: the new exchange paths (wire image, g16 optimizer, sharded optimizer, plain bench --gpus 2) + the comm record
# with and without a high-priority RCCL stream
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_wire.py tests/test_gpu_ddp_single_gpu.py tests/test_gpu_rccl_world1.py tests/test_gpu_bench_contract.py tests/test_gpu_fused_adam.py -x -q > gpurun_out/a_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/a_tests.log
tail -5 gpurun_out/a_tests.log
for hp in 0 1; do
  SZN_RCCL_HIPRI=$hp timeout 600 python bench.py --sub-record comm --steps 10 > gpurun_out/a_comm_hipri$hp.log 2>&1
  grep SUBRECORD gpurun_out/a_comm_hipri$hp.log | cut -c1-1500
done
