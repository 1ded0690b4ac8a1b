"""world_size-2 CPU (gloo) check of the batch-sharded forward: each rank runs its shard of the
global batch and the head collective reassembles exactly the single-process logits."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytorchvideo_amd.parallel import gather_logits, shard_batch, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    torch.set_num_threads(2)
    from pytorchvideo_amd.models import create_x3d
    from oracle.weights import deterministic_fill, seeded_input
    m = create_x3d(input_clip_length=4, input_crop_size=64, model_num_class=10)
    deterministic_fill(m, 0).eval()
    x = seeded_input((global_batch, 3, 4, 64, 64), 3)
    with torch.no_grad():
        local = m(shard_batch(x))
        full = gather_logits(local, global_batch=global_batch)
        want = m(x)
    assert full.shape == want.shape
    assert torch.allclose(full, want, atol=1e-5, rtol=1e-5), (full - want).abs().max()
    torch.save(full, os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def _run(global_batch, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, global_batch, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a, b)  # every rank holds the same gathered logits


def test_equal_shards(tmp_path):
    _run(4, tmp_path)


def test_ragged_shards(tmp_path):
    _run(3, tmp_path)


def test_shard_range_partitions_the_batch():
    for gb in (1, 7, 32, 256):
        for w in (1, 2, 4, 8):
            spans = [shard_range(gb, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
