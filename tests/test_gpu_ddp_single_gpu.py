"""Data parallelism proven on ONE GPU: two ranks share device 0 (gloo backend, CUDA tensors) and each runs the real
engine.TrainStep on its own image; the gradient buckets are issued from the real `layer_done` hooks of the backward pass.
The result must equal a single process stepping on both images (B = 2): same averaged flat gradient, same weights after
the optimizer step (fp32, eval mode = no dropout, deterministic kernels).  The 8-GPU RCCL curve itself is measured by the
driver only (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

E, K, H = 20, 33, 64


def _data():
    from zeroshotsemanticsegmentation_amd import synth
    return synth.make_images(2, H, H, seed=61), synth.make_labels(2, H, H, K, seed=62, block=16), synth.make_embeddings(K, E)


def _worker(rank, world, port, optname, comm, q, arch="FCN32s"):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from zeroshotsemanticsegmentation_amd import engine, models
        x, t, emb = _data()
        dev = torch.device("cuda", 0)
        m = getattr(models, arch)(E)
        m.load_synthetic(1337, device=dev)
        m.eval()
        ts = engine.TrainStep(m, emb, optimizer=optname, lr=1e-5, precision=torch.float32, fused_head=True, bucket_mb=25,
                              grad_comm_dtype=torch.bfloat16 if comm == "bf16" else torch.float32)
        assert ts.world == 2 and len(ts.buckets.buckets) >= 4
        issued = []
        orig = ts.buckets.layer_done

        def spy(name):
            issued.append((name, len(ts.buckets.works)))
            orig(name)
        ts.buckets.layer_done = spy
        loss, _ = ts.step(torch.from_numpy(x[rank:rank + 1]).to(dev), torch.from_numpy(t[rank:rank + 1]).to(dev))
        torch.cuda.synchronize()
        out = {"rank": rank, "loss": float(loss), "layers_reported": [n for n, _ in issued]}
        if rank == 0:
            out["gw"] = (ts.flat_gw * 0.5).cpu().numpy()          # what the optimizer consumed: sum x 1/world
            out["gb"] = (ts.flat_gb * 0.5).cpu().numpy()
            out["w"] = ts.flat_w.cpu().numpy()
            out["b"] = ts.flat_b.cpu().numpy()
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as ex:      # surface the failure instead of a queue timeout
        import traceback
        q.put({"rank": rank, "error": "%r\n%s" % (ex, traceback.format_exc())})


@pytest.mark.parametrize("optname,comm,arch", [("adam", "fp32", "FCN32s"), ("sgd", "fp32", "FCN32s"), ("adam", "bf16", "FCN32s"),
                                               ("sgd", "fp32", "FCN8s")])
def test_two_ranks_on_one_gpu_equal_one_process_batch2(optname, comm, arch):
    from zeroshotsemanticsegmentation_amd import engine, models
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, optname, comm, q, arch)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        o = q.get(timeout=600)
        assert "error" not in o, o.get("error")
        res[o["rank"]] = o
    for p in procs:
        p.join(120)
    # backward reports every optimizer-visible layer, last layer first
    assert res[0]["layers_reported"][0] == "score_fr" and res[0]["layers_reported"][-1] == "conv1_1"
    assert set(res[0]["layers_reported"]) == set(models._OPT_LAYERS8 if arch == "FCN8s" else models._OPT_LAYERS)
    # the single-process reference: both images in one batch
    x, t, emb = _data()
    m = getattr(models, arch)(E)
    m.load_synthetic(1337, device=torch.device("cuda", 0))
    m.eval()
    ts = engine.TrainStep(m, emb, optimizer=optname, lr=1e-5, precision=torch.float32, fused_head=True)
    loss, _ = ts.step(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - 0.5 * (res[0]["loss"] + res[1]["loss"])) < 1e-6
    gw, gb = ts.flat_gw.cpu().numpy(), ts.flat_gb.cpu().numpy()
    tol = 1e-5 if comm == "fp32" else 1e-2            # bf16 wire format: 2^-9 relative per rank
    assert np.abs(res[0]["gw"] - gw).max() < tol * np.abs(gw).max()
    assert np.abs(res[0]["gb"] - gb).max() < 10 * tol * np.abs(gb).max()
    if comm == "fp32":
        w, b = ts.flat_w.cpu().numpy(), ts.flat_b.cpu().numpy()
        if optname == "sgd":        # linear in the gradient: tight
            assert np.abs(res[0]["w"] - w).max() < 1e-8 + 1e-5 * 1e-5 * np.abs(gw).max()
        else:                       # Adam's first step is lr * g / (|g| + eps): compare where the gradient is well above eps
            big = np.abs(gw) > 1e-6
            assert big.mean() > 0.5
            assert np.abs(res[0]["w"] - w)[big].max() < 2e-7
        assert np.abs(res[0]["b"] - b).max() < 1e-6
