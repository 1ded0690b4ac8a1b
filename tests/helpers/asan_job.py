"""Host-side workout of the C ABI for the AddressSanitizer build (run by tests/test_sanitizers.py in a python started
with LD_PRELOAD=libclang_rt.asan and PV_MI355X_LIB=<asan build>): launch plans (add / size checks / destroy / joint
handles), the knob table from two threads, the communicator over the librccl double, then the random-descriptor fuzz of
tests/helpers/abi_fuzz.py.  No GPU: every compute entry point must come back with a status, and ASan must stay silent."""
import ctypes as C
import os
import runpy
import sys
import threading

from pytorchvideo_amd import _lib as L

lib = L.lib()
assert "asan" in L.LIB_PATH, L.LIB_PATH

# plans: every descriptor kind, wrong sizes, out-of-range queries
plans = []
for rep in range(50):
    p = C.c_void_p(lib.pv_plan_create())
    for kind, cls in L.DESC_FOR_OP.items():
        d = cls()
        assert lib.pv_plan_add(p, kind, C.byref(d), C.sizeof(d)) >= 0
        assert lib.pv_plan_add(p, kind, C.byref(d), C.sizeof(d) - 4) == L.PV_ERR_INVALID
    assert lib.pv_plan_add(p, 99, C.byref(L.Conv3dDesc()), 8) == L.PV_ERR_INVALID
    assert lib.pv_plan_size(p) == len(L.DESC_FOR_OP)
    assert lib.pv_plan_op_kernel(p, -1) == b"" and lib.pv_plan_op_kernel(p, 10 ** 6) == b""
    assert lib.pv_plan_launch_range(p, 5, 2, None) == L.PV_ERR_INVALID
    assert lib.pv_plan_launch(p, None) < 0            # no GPU (or an all-zero descriptor): a status, not a crash
    assert lib.pv_plan_graph_launch(p, None) == L.PV_ERR_INVALID
    plans.append(p)
j = C.c_void_p(lib.pv_joint_create())
arr = (C.c_void_p * 2)(plans[0], None)
assert lib.pv_joint_build(j, arr, 2) == L.PV_ERR_INVALID and lib.pv_joint_launch(j, None) == L.PV_ERR_INVALID
assert lib.pv_joint_build(j, arr, 17) == L.PV_ERR_INVALID
lib.pv_joint_destroy(j)
for p in plans:
    lib.pv_plan_destroy(p)
lib.pv_plan_destroy(None)

# the knob table is shared by every thread of the process
def hammer(seed):
    for i in range(2000):
        lib.pv_tune_set(("k%d" % ((seed + i) % 37)).encode(), i)
        if i % 500 == 499:
            lib.pv_tune_clear()
ts = [threading.Thread(target=hammer, args=(s,)) for s in range(4)]
[t.start() for t in ts]
[t.join() for t in ts]
lib.pv_tune_clear()
assert lib.pv_tune_set(b"x" * 60, 1) == L.PV_ERR_INVALID

# communicator over the librccl double (host buffers)
stub = os.environ.get("PV_RCCL_LIB")
if stub:
    import torch
    from pytorchvideo_amd.parallel import HeadComm
    for _ in range(3):
        comm = HeadComm(lib_paths=stub)
        x, out = torch.arange(4000, dtype=torch.float32), torch.zeros(4000)
        comm.all_gather(x, out)
        assert torch.equal(x, out)
        src = (L.GatherSrc * 1)(L.GatherSrc(x.data_ptr(), 400, 400, 10))
        assert lib.pv_forward_gather(None, None, comm.handle, src, 1, None, out.data_ptr(), None) == L.PV_ERR_INVALID
        comm.close()
    assert lib.pv_comm_probe(b"/nonexistent.so") == L.PV_ERR_HIP

sys.argv = ["abi_fuzz.py", "11", "60"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "abi_fuzz.py"), run_name="__main__")
print("asan job ok")
