"""Which XCC / SE / CU the workgroups of a CU-masked stream land on (hipExtStreamCreateWithCUMask bit layout):
python devtools/tools_probe_cumask.py"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import ctypes as C, collections
import cuda_qr_amd as q
lib = q.lib
lib.qrd_probe_cumask.argtypes = [C.POINTER(C.c_uint), C.c_int, C.c_int, C.POINTER(C.c_uint)]
lib.qrd_probe_cumask.restype = C.c_int
q.check(lib.qrd_init(), "init")
def probe(bits, label):
    words = (C.c_uint * 8)(*([0] * 8))
    for b in bits: words[b >> 5] |= 1 << (b & 31)
    n = 4096
    out = (C.c_uint * n)()
    rc = lib.qrd_probe_cumask(words, 8, n, out)
    assert rc == 0, rc
    per = collections.Counter()
    cus = collections.defaultdict(set)
    for v in out:
        xcc = v & 0xf; hw = v >> 8
        cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
        per[xcc] += 1; cus[xcc].add((se, sh, cu))
    print("%-28s XCC -> #WGs: %s" % (label, dict(sorted(per.items()))))
    print("%-28s XCC -> distinct (se,sh,cu): %s" % ("", {k: len(v) for k, v in sorted(cus.items())}))
probe(range(0, 32), "bits 0..31")
probe(range(32, 64), "bits 32..63")
probe(range(0, 8), "bits 0..7")
probe(range(0, 64), "bits 0..63")
probe(range(0, 48), "bits 0..47")
probe([b for b in range(256) if (b & 31) < 4], "bits (b&31)<4")
probe(range(0, 256, 8), "bits 0,8,16,...")
probe(range(0, 256), "all 256")
