"""CPU, world_size 2, gloo: the data-parallel gradient exchange (engine.GradBuckets) used by TrainStep over RCCL."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zeroshotsemanticsegmentation_amd.engine import GradBuckets
    layers, off = [], 0
    for i, n in enumerate([27, 5000, 120, 9000, 64, 7]):            # forward-order layer sizes
        layers.append(("L%d" % i, off, n))
        off += n
    flat = torch.arange(off, dtype=torch.float32) * (rank + 1)
    bias = torch.full((11,), float(rank + 1))
    gb = GradBuckets(flat, layers, bucket_elems=4000, extra=[bias])
    # buckets are contiguous, cover everything once, in backward order, each >= 4000 elems except the last
    spans = [(o, e) for o, e, _ in gb.buckets]
    assert spans[0][1] == off and spans[-1][0] == 0
    assert all(spans[i][0] == spans[i + 1][1] for i in range(len(spans) - 1))
    assert all(e - o >= 4000 for o, e in spans[:-1])
    for name, _, _ in reversed(layers):                               # backward completion order
        gb.layer_done(name)
    gb.finish()
    want = torch.arange(off, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(flat, want) and torch.equal(bias, torch.full((11,), float(sum(r + 1 for r in range(world)))))
    # the optimizer applies grad_scale = 1/world: mean of the rank gradients
    ok = ok and torch.allclose(flat / world, torch.arange(off, dtype=torch.float32) * 1.5)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_grad_buckets_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _worker2(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zeroshotsemanticsegmentation_amd.engine import GradBuckets, allreduce_param_grads
    # bf16 wire format: sums agree with the fp32 sum to bf16 rounding, staging buffer leaves `flat` fp32
    layers = [("a", 0, 3000), ("b", 3000, 5000)]
    g = torch.Generator().manual_seed(7 + rank)
    flat = torch.randn(8000, generator=g)
    mine = flat.clone()
    gb = GradBuckets(flat, layers, bucket_elems=4000, comm_dtype=torch.bfloat16)
    for name in ("b", "a"):
        gb.layer_done(name)
    gb.finish()
    others = [torch.randn(8000, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
    want = sum(o.bfloat16().float() for o in others)
    ok = flat.dtype == torch.float32 and torch.allclose(flat, want, rtol=1e-2, atol=1e-2) and not torch.equal(flat, mine)
    # explicit parameter list (autograd trainer paths): mean of the rank gradients, parameters without a gradient skipped
    ps = [torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2))]
    ps[0].grad = torch.full((4, 3), float(rank + 1))
    ps[1].grad = torch.arange(5, dtype=torch.float32) * (rank + 1)
    allreduce_param_grads(ps)
    ok = ok and torch.equal(ps[0].grad, torch.full((4, 3), 1.5)) and torch.equal(ps[1].grad, torch.arange(5, dtype=torch.float32) * 1.5)
    ok = ok and ps[2].grad is None
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bf16_wire_and_param_list_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker2, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_single_process_is_noop():
    from zeroshotsemanticsegmentation_amd.engine import GradBuckets
    flat = torch.ones(10)
    gb = GradBuckets(flat, [("a", 0, 4), ("b", 4, 6)], bucket_elems=1)
    gb.layer_done("b"); gb.layer_done("a"); gb.finish()
    assert torch.equal(flat, torch.ones(10)) and gb.world == 1


def _worker_sharded(rank, world, port, q):
    """reduce-scatter + rank-sharded update + all-gather == all-reduce + replicated update, bit for bit (two ranks: the sum of two
    terms does not depend on the order), for the fp32 wire, the staged bf16 wire and the direct bf16 wire; exchange off = untouched"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zeroshotsemanticsegmentation_amd.engine import GradBuckets
    layers, off = [], 0
    for i, n in enumerate([64, 4096, 128, 8192, 64, 32]):             # forward order; every bucket splits into 2 x whole 16-B groups
        layers.append(("L%d" % i, off, n))
        off += n
    grads = [torch.randn(off, generator=torch.Generator().manual_seed(11 + r)) for r in range(world)]
    w0 = torch.randn(off, generator=torch.Generator().manual_seed(5))

    def run(sharded, comm, direct):
        flat = grads[rank].clone()
        gb = GradBuckets(flat, layers, bucket_elems=4000, comm_dtype=comm, sharded=sharded, direct=direct)
        if gb.direct:                                                 # the kernels write the wire image themselves
            gb.stage.copy_(flat)
        for name, _, _ in reversed(layers):
            gb.layer_done(name)
        gb.finish()
        g = gb.stage.float() if gb.direct else flat                   # what the optimizer reads
        w = w0.clone()
        if sharded:
            for o, e, _ in gb.buckets:
                lo, hi = gb.shard(o, e)
                w[lo:hi] -= 0.1 * g[lo:hi] / world                    # this rank's slice only
            for wk in gb.gather_weights(w):
                wk.wait()
        else:
            w -= 0.1 * g / world
        return w, gb

    ok = True
    for comm, direct in ((torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)):
        wa, ga = run(False, comm, direct)
        ws, gs = run(True, comm, direct)
        ok = ok and torch.equal(wa, ws) and gs.sharded and not ga.sharded
        ok = ok and gs.issued == 2 * len(gs.buckets) and ga.issued == len(ga.buckets)
    # the all-gathers waited for ONE BUCKET AT A TIME in forward order (what the next forward pass of engine.TrainStep does through
    # wait_weights): after bucket k's wait its range holds the replicated result, whatever the later buckets are doing
    wa, _ = run(False, torch.float32, False)
    flat = grads[rank].clone()
    gb = GradBuckets(flat, layers, bucket_elems=4000, sharded=True)
    for name, _, _ in reversed(layers):
        gb.layer_done(name)
    gb.finish()
    w = w0.clone()
    for o, e, _ in gb.buckets:
        lo, hi = gb.shard(o, e)
        w[lo:hi] -= 0.1 * flat[lo:hi] / world
    pend = gb.gather_weights(w, spans=True)
    ok = ok and [sp for sp, _ in pend] == sorted(sp for sp, _ in pend) and pend[0][0][0] == 0 and len(pend) == len(gb.buckets)
    for (o, e), wk in pend:
        wk.wait()
        ok = ok and torch.equal(w[o:e], wa[o:e])
    ok = ok and torch.equal(w, wa)
    exact = w0 - 0.1 * sum(grads) / world
    wa, _ = run(False, torch.float32, False)
    ok = ok and torch.allclose(wa, exact, rtol=0, atol=1e-6)
    # exchange off: nothing is issued, the gradient stays this rank's
    flat = grads[rank].clone()
    gb = GradBuckets(flat, layers, bucket_elems=4000, enabled=False)
    for name, _, _ in reversed(layers):
        gb.layer_done(name)
    gb.finish()
    ok = ok and not gb.active and gb.issued == 0 and torch.equal(flat, grads[rank])
    # a bucket that does not split into whole 16-B groups per rank is refused
    try:
        GradBuckets(torch.zeros(10), [("a", 0, 10)], bucket_elems=4, sharded=True)
        ok = False
    except Exception:
        pass
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_optimizer_equals_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 25500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _worker_world8(rank, world, port, q):
    """the bucket / shard arithmetic at the rank count of BASELINE configs[3] (eight ranks; CPU tensors over gloo): every wire mode, all-reduce and
    reduce-scatter + sharded update + bucket-by-bucket all-gather, against the exact mean gradient -- with eight terms the order of the sum matters, so
    the comparison is to fp32 rounding (fp32 wire) / one bf16 rounding per rank (16-bit wires), and all ranks must end with IDENTICAL weights"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zeroshotsemanticsegmentation_amd.engine import GradBuckets
    layers, off = [], 0
    for i, n in enumerate([64, 4096, 128, 8192, 64, 32]):             # forward order; every bucket splits into 8 x whole 16-B groups
        layers.append(("L%d" % i, off, n))
        off += n
    grads = [torch.randn(off, generator=torch.Generator().manual_seed(11 + r)) for r in range(world)]
    w0 = torch.randn(off, generator=torch.Generator().manual_seed(5))
    exact = w0 - 0.1 * sum(g.double() for g in grads).float() / world
    ok = True
    for comm, direct, sharded, tol in ((torch.float32, False, False, 1e-6), (torch.float32, False, True, 1e-6), (torch.bfloat16, False, False, 4e-3),
                                       (torch.bfloat16, True, False, 4e-3), (torch.bfloat16, True, True, 4e-3)):
        flat = grads[rank].clone()
        gb = GradBuckets(flat, layers, bucket_elems=4000, comm_dtype=comm, sharded=sharded, direct=direct)
        ok = ok and gb.world == 8 and gb.active
        if gb.direct:
            gb.stage.copy_(flat)
        for name, _, _ in reversed(layers):
            gb.layer_done(name)
        gb.finish()
        g = gb.stage.float() if gb.direct else flat
        w = w0.clone()
        if sharded:
            for o, e, _ in gb.buckets:
                lo, hi = gb.shard(o, e)
                ok = ok and (hi - lo) * world == e - o and (hi - lo) % 4 == 0
                w[lo:hi] -= 0.1 * g[lo:hi] / world
            for (o, e), wk in gb.gather_weights(w, spans=True):
                wk.wait()
        else:
            w -= 0.1 * g / world
        ok = ok and float((w - exact).abs().max()) < tol
        # every rank holds the same weights afterwards (rank 0's copy is broadcast and compared)
        ref = w.clone()
        dist.broadcast(ref, 0)
        ok = ok and torch.equal(ref, w)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucket_and_shard_arithmetic_at_eight_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_world8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(8)]
