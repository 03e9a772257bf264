"""Latency of ONE TSQR step on one GPU (cuda_qr_amd.tsqr.rank_step_latency): local QR alone, and the complete step of a rank of a
P-GPU run with the collective replaced by device copies of its own factor.  MI355XQR_TSQR_PIPE=0|1: one collective after the local QR /
the panel-pipelined exchange.   python devtools/tools_tsqr_latency.py m_local x n x P [x nb] ..."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import json, sys
import cuda_qr_amd as q
from cuda_qr_amd import tsqr as T

for spec in sys.argv[1:]:
    f = [int(x) for x in spec.split("x")]
    out = T.rank_step_latency(q, f[0], f[1], f[2], f[3] if len(f) > 3 else 128)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)
