"""Single-rank RCCL round trip through the library's OWN loader (qr_comm.hip: dlopen librccl, ncclGetUniqueId, ncclCommInitRank,
ncclAllGather on a plan's stream) in a process that has torch -- and torch's bundled HIP runtime and RCCL -- loaded, as bench.py has.
What a one-GPU box can check of the multi-GPU transport: library resolution, symbol binding, the bootstrap, a collective kernel on the
library's stream.   python tests/rccl_single_rank_roundtrip.py"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import ctypes as C, json
import torch
import cuda_qr_amd as q

lib = q.lib
uid = (C.c_ubyte * 128)()
q.check(lib.qrd_comm_unique_id(uid), "unique id")
comm = C.c_void_p()
lib.qrd_comm_init_rank.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int]
q.check(lib.qrd_comm_init_rank(C.byref(comm), 1, uid, 0), "comm init rank")
n = C.c_int(-1)
lib.qrd_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
q.check(lib.qrd_comm_count(comm, C.byref(n)), "comm count")
p = q.Plan(1024, 64)
cnt = 512 * 128
lib.qrd_allgather_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
# (a) torch-allocated buffers (torch's HIP runtime), (b) the library's own allocations (its HIP runtime)
send = torch.arange(cnt, dtype=torch.float64, device="cuda") * 0.5
recv = torch.zeros(cnt, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
q.check(lib.qrd_allgather_f64(comm, p.stream, send.data_ptr(), recv.data_ptr(), cnt), "all-gather (torch buffers)")
p.sync()
ok_a = bool(torch.equal(send, recv))
d0, d1 = C.c_void_p(), C.c_void_p()
lib.qr_device_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
q.check(lib.qr_device_malloc(C.byref(d0), 8 * cnt)); q.check(lib.qr_device_malloc(C.byref(d1), 8 * cnt))
host = (torch.arange(cnt, dtype=torch.float64) * 0.25).contiguous()
lib.qr_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.qr_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
q.check(lib.qr_copy_to_device(d0, host.data_ptr(), 8 * cnt))
q.check(lib.qrd_allgather_f64(comm, p.stream, d0, d1, cnt), "all-gather (library buffers)")
p.sync()
back = torch.empty(cnt, dtype=torch.float64)
q.check(lib.qr_copy_to_host(back.data_ptr(), d1, 8 * cnt))
ok_b = bool(torch.equal(back, host))
lib.qr_device_free.argtypes = [C.c_void_p]
lib.qr_device_free(d0); lib.qr_device_free(d1)
p.close()
lib.qrd_comm_destroy.argtypes = [C.c_void_p]
rc = lib.qrd_comm_destroy(comm)
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if ("rccl" in l or "amdhip64" in l or "hsa-runtime" in l) and "/" in l})
print(json.dumps({"ranks_seen_by_rccl": n.value, "allgather_torch_buffers_ok": ok_a, "allgather_library_buffers_ok": ok_b,
                  "destroy_rc": rc, "mapped_runtime_libraries": maps}))
_sys.exit(0 if (n.value == 1 and ok_a and ok_b and rc == 0) else 1)
