"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding + the one all-gather of
fixed-shape proposals (the compute is a deterministic stand-in: the HIP ops need a GPU)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_every_batch():
    from ws3d_amd.dist import shard_range
    for gb in (1, 2, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(gb, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from ws3d_amd import dist as wd
    world, rank, local = wd.init("gloo")
    GB, K = int(os.environ["GB"]), 5

    def compute(s, e):
        b = e - s
        scene = torch.arange(s, e, dtype=torch.float32)
        boxes = scene.view(b, 1, 1).expand(b, K, 7) * 10 + torch.arange(7, dtype=torch.float32)
        scores = scene.view(b, 1).expand(b, K) + torch.arange(K, dtype=torch.float32) / 100
        count = (torch.arange(s, e) %% (K + 1)).to(torch.int64)
        return boxes.contiguous(), scores.contiguous(), count

    packed, count = wd.run_sharded(GB, compute)
    assert packed.shape == (GB, K, 8) and count.shape == (GB,)
    full_b, full_s, full_c = compute(0, GB)
    assert torch.equal(packed[:, :, :7], full_b) and torch.equal(packed[:, :, 7], full_s)
    assert torch.equal(count, full_c)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(os.environ["OUT"], "rank%%d.ok" %% rank), "w").write("ok")
""")


@pytest.mark.parametrize("gb", [8, 5])
def test_two_process_gloo_all_gather(tmp_path, gb):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, GB=str(gb), MASTER_ADDR="127.0.0.1", OUT=str(tmp_path))
    cmd = [sys.executable, "-B", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + gb), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


EXCHANGE_WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from ws3d_amd import dist as wd
    world, rank, local = wd.init("gloo")
    GB, K, STEPS = int(os.environ["GB"]), 5, 4
    s, e = wd.shard_range(GB, world, rank)
    b = e - s
    calls = {"n": 0}
    real = dist.all_gather_into_tensor

    def counted(*a, **kw):
        calls["n"] += 1
        return real(*a, **kw)

    dist.all_gather_into_tensor = counted
    ex = wd.ProposalExchange(b, K, GB, "cpu")
    send_ptr, recv_ptr = ex.send.data_ptr(), ex.recv.data_ptr()
    for step in range(STEPS):
        scene = torch.arange(s, e, dtype=torch.float32) + 100 * step
        packed = scene.view(b, 1, 1).expand(b, K, 8) * 10 + torch.arange(8, dtype=torch.float32)
        count = (torch.arange(s, e) + step) %% (K + 1)
        ex.fill(packed, count)                    # (on the GPU the selection kernel writes ex.send itself: ws3d_select_proposals_send)
        got, cnt = ex.gather()
        # exactly ONE collective per step, on the SAME two buffers every step
        assert calls["n"] == step + 1 == ex.collectives
        assert ex.send.data_ptr() == send_ptr and ex.recv.data_ptr() == recv_ptr
        if GB %% world == 0:                       # even shards: the results are VIEWS of the receive buffer, nothing is allocated
            assert got.data_ptr() == recv_ptr and cnt.untyped_storage().data_ptr() == ex.recv.untyped_storage().data_ptr()
        full = (torch.arange(GB, dtype=torch.float32) + 100 * step).view(GB, 1, 1).expand(GB, K, 8) * 10 + torch.arange(8, dtype=torch.float32)
        assert got.shape == (GB, K, 8) and torch.equal(got, full)
        assert torch.equal(cnt, ((torch.arange(GB) + step) %% (K + 1)).to(torch.float32))
        # ... and equal to the allocate-per-step form of rounds 1-5
        want, wcnt = wd.all_gather_proposals(packed.contiguous(), count, GB)
        calls["n"] -= 1
        assert torch.equal(want, got) and torch.equal(wcnt.to(torch.float32), cnt)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(os.environ["OUT"], "rank%%d.ok" %% rank), "w").write("ok")
""")


@pytest.mark.parametrize("gb", [8, 5])
def test_proposal_exchange_is_one_collective_on_resident_buffers(tmp_path, gb):
    """the step's exchange allocates nothing and issues exactly one collective (world 2, gloo, even and
    uneven shards); its results equal all_gather_proposals' (the allocate-per-step form)"""
    script = tmp_path / "worker.py"
    script.write_text(EXCHANGE_WORKER % ROOT)
    env = dict(os.environ, GB=str(gb), MASTER_ADDR="127.0.0.1", OUT=str(tmp_path))
    cmd = [sys.executable, "-B", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29620 + gb), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_more_rccl_ranks_than_devices_raises_instead_of_wrapping(monkeypatch):
    """dist.init over RCCL never puts two ranks on one device: the device pick raises; only the
    explicit gloo test mode wraps LOCAL_RANK onto the devices there are"""
    import pytest
    import torch
    from ws3d_amd import dist as wdist
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    # a LOCAL_RANK beyond the node's devices, or more ranks ON THIS NODE than devices (LOCAL_WORLD_SIZE, as torchrun exports it)
    for world, local, lws in ((4, 3, None), (3, 2, None), (4, 1, 4), (3, 0, 3)):
        with pytest.raises(RuntimeError, match="one device per rank"):
            wdist._local_device(local, world, "nccl", lws)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")            # ... read from the environment when not passed
    with pytest.raises(RuntimeError, match="one device per rank"):
        wdist._local_device(1, 4, "nccl")
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert [wdist._local_device(l, 2, "nccl", 2) for l in (0, 1)] == [0, 1]
    assert [wdist._local_device(l, 4, "gloo", 4) for l in range(4)] == [0, 1, 0, 1]
    # multi-node jobs: the GLOBAL world size says nothing about this node: 2 nodes x 8 GPUs
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert wdist._local_device(3, 16, "nccl", 8) == 3
    assert [wdist._local_device(l, 16, "nccl") for l in range(8)] == list(range(8))
    with pytest.raises(RuntimeError, match="one device per rank"):
        wdist._local_device(8, 16, "nccl", 8)


def test_rccl_without_a_device_is_an_error(monkeypatch):
    import pytest
    import torch
    from ws3d_amd import dist as wdist
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    for k, v in (("WORLD_SIZE", "2"), ("RANK", "0"), ("LOCAL_RANK", "0")):
        monkeypatch.setenv(k, v)
    with pytest.raises(RuntimeError, match="without a HIP device"):
        wdist.init("nccl")
