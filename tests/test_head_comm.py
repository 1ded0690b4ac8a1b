"""The head collective driven from C (include/pv_mi355x.h `pv_comm_*`).

CPU: world-size-2 processes bootstrap a communicator exactly as `bench.py --gpus N` does (rank 0 draws the id, the gloo
group carries it, every rank calls pv_comm_create) and all-gather their logits through `pv_comm_all_gather`, with a
test double of librccl (tests/helpers/rccl_stub.c: host buffers through shared memory) behind the dlopen.
GPU: the REAL librccl on the MI355X with a one-rank communicator, and pv_forward_gather (graph replay + row
collection + all-gather as one C call) against the ordinary forward."""
import ctypes as C
import os
import socket
import subprocess

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build_stub(tmp_path):
    so = os.path.join(str(tmp_path), "librccl_stub.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "helpers", "rccl_stub.c"), "-lrt"])
    return so


def _worker(rank, world, port, stub, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PV_RCCL_LIB"] = stub
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorchvideo_amd.parallel import HeadComm, gather_logits
    comm = HeadComm()
    assert comm.rank == rank and comm.world_size == world and comm.library == stub
    for step in range(3):    # the communicator is reused step after step
        g = torch.Generator().manual_seed(100 * step + rank)
        local = torch.randn(4, 400, generator=g)
        recv = torch.empty(world * 4, 400)
        comm.all_gather(local, recv)
        want = gather_logits(local, global_batch=4 * world)     # the torch.distributed route
        assert torch.equal(recv, want)
    torch.save(recv, os.path.join(out_dir, "r%d.pt" % rank))
    comm.close()
    dist.destroy_process_group()


def test_two_ranks_all_gather_through_the_c_communicator(tmp_path):
    stub = _build_stub(tmp_path)
    mp.spawn(_worker, args=(2, _free_port(), stub, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a, b)


def test_single_rank_and_error_paths(tmp_path):
    from pytorchvideo_amd import _lib as L
    from pytorchvideo_amd.parallel import HeadComm
    stub = _build_stub(tmp_path)
    comm = HeadComm(lib_paths=stub)
    x = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    out = torch.zeros(3, 4)
    comm.all_gather(x, out)
    assert torch.equal(out, x)
    lib = L.lib()
    assert lib.pv_comm_rank(comm.handle) == 0 and lib.pv_comm_world(comm.handle) == 1
    # descriptor errors are statuses, not crashes
    assert lib.pv_comm_all_gather(None, x.data_ptr(), out.data_ptr(), 48, None) == L.PV_ERR_INVALID
    h = C.c_void_p()
    ident = C.create_string_buffer(128)
    assert lib.pv_comm_create(C.byref(h), ident, 2, 2, stub.encode()) == L.PV_ERR_INVALID      # rank >= world
    assert lib.pv_comm_unique_id(ident, b"/nonexistent/librccl.so") == L.PV_ERR_HIP
    assert b"no RCCL library" in lib.pv_last_error()
    assert lib.pv_forward_gather(None, None, None, None, 0, None, None, None) == L.PV_ERR_INVALID
    comm.close()


@pytest.mark.gpu
def test_real_rccl_one_rank_and_forward_gather_on_the_gpu():
    from oracle.weights import deterministic_fill, seeded_input
    from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
    from pytorchvideo_amd.models import create_x3d
    from pytorchvideo_amd.parallel import HeadComm, ShardedForward
    comm = HeadComm()                      # the real librccl (torch's copy), one rank
    assert "rccl" in comm.library
    send = torch.randn(8, 400, device="cuda")
    recv = torch.zeros(8, 400, device="cuda")
    comm.all_gather(send, recv)
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    for streams in (1, 2):
        m = create_x3d(input_clip_length=4, input_crop_size=64, model_num_class=10)
        deterministic_fill(m, 0).eval()
        x = seeded_input((4, 3, 4, 64, 64), 3).cuda().bfloat16()
        transmute_model(m, "mi355x")
        dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=streams)
        want = dm(x).clone()
        for c in (comm, None):
            sf = ShardedForward(dm, c)
            for _ in range(2):
                got = sf(x)
            torch.cuda.synchronize()
            assert got.shape == want.shape and torch.equal(got, want)
    comm.close()


def test_an_rccl_of_an_unknown_major_version_is_refused_and_the_binding_is_cached(tmp_path):
    """the NCCL call signatures are declared locally, so a library that reports another major version is
    refused (never called through a guessed ABI); and a binding is made once per library list, not once per call."""
    from pytorchvideo_amd import _lib as L
    stub = _build_stub(tmp_path)
    lib = L.lib()
    assert lib.pv_comm_probe(stub.encode()) == 0
    # the same list again: served from the cache (the environment is not even consulted any more)
    os.environ["PV_RCCL_STUB_VERSION"] = "30000"
    try:
        assert lib.pv_comm_probe(stub.encode()) == 0
        # a different list naming the same file binds afresh and now sees major version 3
        rc = lib.pv_comm_probe((stub + ":" + stub).encode())
        assert rc != 0
        assert b"version" in (lib.pv_last_error() or b"")
    finally:
        del os.environ["PV_RCCL_STUB_VERSION"]
