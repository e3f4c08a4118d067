"""World-size-2 CPU (gloo) test of the multi-GPU layer: utterance partitioning, the one-shot weight-arena
broadcast and the gather of generated ids.  (The engine itself needs a GPU; here the arena bytes travel
through the same code path on CPU tensors.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model_dir, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qwen3_asr_rs_amd import distributed as D
    arena = D.broadcast_arena(model_dir, torch.device("cpu"), src=0)
    digest = int(arena.to(torch.int64).sum().item())
    n_total = 5
    s, e = D.partition(n_total, world, rank)
    local = [[100 * i + k for k in range(i + 1)] for i in range(s, e)]
    allids = D.gather_ids(local, n_total)
    q.put((rank, digest, arena.numel(), (s, e), allids))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_properties():
    sys.path.insert(0, ROOT)
    from qwen3_asr_rs_amd.distributed import partition
    for n in (0, 1, 5, 32, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [partition(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert partition(256, 8, 3) == (96, 128)  # config 4: 32 clips per GPU


def test_arena_broadcast_and_gather_world2(tiny_dir, lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tiny_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    (r0, d0, n0, s0, ids0), (r1, d1, n1, s1, ids1) = res
    assert d0 == d1 and n0 == n1 and n0 > 0          # identical arena bytes on both ranks
    assert s0 == (0, 3) and s1 == (3, 5)
    expect = [[100 * i + k for k in range(i + 1)] for i in range(5)]
    assert ids0 == expect and ids1 == expect


def test_native_group_partition_matches_python(lib):
    """q3a_group_partition (csrc/group.cpp) is the split q3a_group_transcribe uses: identical to distributed.partition."""
    import ctypes as C
    from qwen3_asr_rs_amd.distributed import partition
    for n in (1, 2, 5, 32, 33, 256, 257):
        for w in (1, 2, 3, 8):
            covered = []
            for r in range(w):
                b, e = C.c_int32(), C.c_int32()
                lib.q3a_group_partition(n, w, r, C.byref(b), C.byref(e))
                assert (b.value, e.value) == partition(n, w, r)
                covered += list(range(b.value, e.value))
            assert covered == list(range(n))


def test_native_group_fails_loudly_without_gpu(lib, tiny_dir):
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.q3a_group_create(tiny_dir.encode(), 2, None, None, C.byref(h)) != 0
    assert b"no HIP device" in lib.q3a_last_error(None)
