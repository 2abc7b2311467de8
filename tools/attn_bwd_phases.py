"""Where one workgroup of the attention backward spends an item (shader-clock stamps).  Needs a DEBUG build of the library --
the stamp checks cost the kernel 6 %, so the default build compiles them out:
    make -C mmlearn_amd/csrc clean && make -C mmlearn_amd/csrc EXTRA="-DMMK_DEBUG_SWITCHES -DMMK_ATTN_STAMPS_BUILD"
    MMK_ATTN_STAMPS=1 python tools/attn_bwd_phases.py        (B=1024 H=12 L=197 by default; env B, L, P)
    make -C mmlearn_amd/csrc clean && make -C mmlearn_amd/csrc"""
import ctypes as C, json, os, sys
import torch
os.environ.setdefault("MMK_ATTN_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import _lib, kernels as K
B, H, L = int(os.environ.get("B", 1024)), 12, int(os.environ.get("L", 197))
p = float(os.environ.get("P", 0.0))
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(B, L, H * 64, device=dev).bfloat16().view(B, L, H, 64).transpose(1, 2) for _ in range(3))
do = torch.randn(B, L, H, 64, device=dev).bfloat16()
o, lse = K.attn_fwd(q, k, v, 0.125, p, 7)
for _ in range(3):
    K.attn_bwd(q, k, v, o, lse, do, 0.125, p, 7)
buf = (C.c_uint64 * (32 * 16))()
_lib.check(_lib.lib().mmk_attn_debug_stamps(C.cast(buf, C.c_void_p), 32 * 16))
nt = (L + 31) // 32
n_st = 5 + nt   # 0 start, 1 issued, 2 landed, 3 .. 3+nt step barriers, 4+nt stores issued
items = [[buf[i * 16 + k] for k in range(16)] for i in range(32)]
items = [it for it in items if it[0] and it[n_st - 1] > it[0]][2:]   # skip the first two (cold)
def avg(f):
    return sum(f(it) for it in items) / len(items)
out = {"L": L, "items": len(items), "cycles": {
    "issue_loads": avg(lambda t: t[1] - t[0]), "wait_loads": avg(lambda t: t[2] - t[1]),
    "steps": [avg(lambda t, s=s: t[3 + s] - t[2 + s]) for s in range(nt + 1)],
    "final_stores_issue": avg(lambda t: t[4 + nt] - t[3 + nt]),
    "item_total": avg(lambda t: t[4 + nt] - t[0])}}
if nt > 3:   # inside step 3 of key wave 0: 12 S / dP MFMAs issued, 13 softmax arithmetic done, 14 dS tile written, 15 dV / dK MFMAs issued
    out["cycles"]["step3_key_wave0"] = {"S_dP_mfma_issue": avg(lambda t: t[12] - t[5]), "softmax_valu": avg(lambda t: t[13] - t[12]),
                                        "dS_write": avg(lambda t: t[14] - t[13]), "dV_dK_mfma_issue": avg(lambda t: t[15] - t[14]),
                                        "barrier_wait": avg(lambda t: t[6] - t[15])}
    out["cycles"]["step3_key_wave0"] = {k: round(v) for k, v in out["cycles"]["step3_key_wave0"].items()}
nxt = [items[i + 1][0] - items[i][4 + nt] for i in range(len(items) - 1)]
out["cycles"]["gap_to_next_item"] = sum(nxt) / max(len(nxt), 1)
out["cycles"] = {k: ([round(x) for x in v] if isinstance(v, list) else (v if isinstance(v, dict) else round(v))) for k, v in out["cycles"].items()}
print(json.dumps(out))
