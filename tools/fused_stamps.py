"""wall_clock64 timeline of the fused small-batch kernel's blocks (debug build, -DCF_TRSM_STAMPS), W = 16."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("cosmology-model-fit_amd")
data = amd.synthetic.pantheon_like(1701, seed=0)
lk = amd.sn_pantheon.PantheonLikelihood(data["z_cmb"], data["z_hel"], data["obs"], chol=data["chol"])
th = amd.synthetic.walkers(lk.bounds, 4096, seed=1)
lk.log_probability(th)
for _ in range(200):
    lk.log_probability(th[:16])
buf = (C.c_uint64 * (256 * 8))()
assert amd._lib.lib().cf_debug_fused_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
t0 = st[:72, 0].min()
r = lambda a: (a - t0) * 10  # ns
print("walker blocks (0..15): start, residuals done, release add done [ns after the first block's entry]")
for b in range(16):
    print(f"  block {b:2d}: {r(st[b, 0]):6d} {r(st[b, 1]):6d} {r(st[b, 2]):6d}")
print("solve blocks: start, poll done, acquire done, loop + exchange done, hand-off done, [last arriver: end]")
for b in list(range(16, 72)):
    s = st[b]
    print(f"  block {b:2d} (unit {b - 16:2d}): {r(s[0]):6d} {r(s[1]):6d} {r(s[2]):6d} {r(s[3]):6d} {r(s[4]):6d}" + (f" {r(s[5]):6d}" if s[5] > s[4] else ""))
