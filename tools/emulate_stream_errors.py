# Where |U_hip - U_float64| comes from: a numpy emulation of the streaming workgroups' float32
# arithmetic (correctly rounded float32 tables, float32 rate products, float32 tau arguments), term by
# term, against the float64 oracle -- no GPU.  This is how round 3's accuracy pass found its targets
# ("Numerics"); the measured device errors are in profiles/r03/parity_errors.txt.
#   python tools/emulate_stream_errors.py [case] [point]      e.g.  league_1e5 ub
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT + "/oracle", ROOT + "/tests"]
import numpy as np

import cases
import dc_oracle as O

f32 = np.float32
name = sys.argv[1] if len(sys.argv) > 1 else "league_1e5"
point = sys.argv[2] if len(sys.argv) > 2 else "ub"
model = O.MODEL_EXTENDED if point == "clip" else O.MODEL_BASIC
fx = cases.fixtures(name)
z = dict(cases.z_points(model, fx))[point]
U, g, aux = O.potential_and_grad(model, fx, z)
att, dfn = aux["attack"], aux["defence"]
ha = np.broadcast_to(aux["home_advantage"], att.shape)
h, a = fx.home_idx, fx.away_idx
x, y = fx.home_goals.astype(float), fx.away_goals.astype(float)
rho = aux["rho"]
eh, ea = att[h] - dfn[a] + ha[h], att[a] - dfn[h]
lh, la = np.exp(eh), np.exp(ea)
clip = model == O.MODEL_EXTENDED
ch, ca = (lh > 15) & clip, (la > 15) & clip
lhc, lac = np.where(ch, 15.0, lh), np.where(ca, 15.0, la)


def tau_log(lh_, la_, rho_, round_t=False):
    out = np.zeros(lh_.shape)
    for m, c in (((x == 0) & (y == 0), -(lh_ * la_)), ((x == 1) & (y == 0), la_), ((x == 0) & (y == 1), lh_),
                 ((x == 1) & (y == 1), -np.ones_like(lh_))):
        t = 1.0 + rho_ * c[m]
        out[m] = np.log(t.astype(f32).astype(np.float64) if round_t else t)
    return out


# float32 tables (correctly rounded; the device's exp2-based ones differ by an ulp or two, which the
# first-order table correction absorbs either way) and their rounding errors
AH, BD, AA = np.exp(att + ha).astype(f32), np.exp(-dfn).astype(f32), np.exp(att).astype(f32)
eAH = np.log(np.exp(att + ha) / AH.astype(np.float64))
eBD = np.log(np.exp(-dfn) / BD.astype(np.float64))
eAA = np.log(np.exp(att) / AA.astype(np.float64))
lh32, la32 = (AH[h] * BD[a]).astype(f32), (AA[a] * BD[h]).astype(f32)
ph, pa = AH[h].astype(np.float64) * BD[a], AA[a].astype(np.float64) * BD[h]  # exact products
lam_exact = -(lhc + lac).sum()
lam_f32 = -(np.where(ch, 15.0, lh32.astype(np.float64)) + np.where(ca, 15.0, la32.astype(np.float64))).sum()
lam_prod = -(np.where(ch, 15.0, ph) + np.where(ca, 15.0, pa)).sum()
first_order = -((np.where(ch, 0.0, ph) * (eAH[h] + eBD[a])).sum() + (np.where(ca, 0.0, pa) * (eAA[a] + eBD[h])).sum())
print(f"{name}/{point}: N={fx.n} U={U:.6f}; clipped fixtures {int(ch.sum())} / {int(ca.sum())}")
print(f"  rate sums, exact float32 products + first-order table correction : {lam_prod + first_order - lam_exact:+.3e}")
print(f"  float32 rounding of the rate PRODUCT, once per pair               : {lam_f32 - lam_prod:+.3e}")
lane = 32.0
slam32 = (lane * (lh32 + la32).astype(f32)).astype(f32).astype(np.float64) / lane
print(f"  nall (lh + la) rounded to float32 per lane                        : {-(slam32.sum()) - (-(lh32.astype(np.float64) + la32).sum()):+.3e}")
print(f"  tau argument 1 + rho c rounded to float32                         : {tau_log(lhc, lac, rho, True).sum() - tau_log(lhc, lac, rho).sum():+.3e}")
if clip:
    sp = -((x[ch] * (eAH[h[ch]] + eBD[a[ch]])).sum() + (y[ca] * (eAA[a[ca]] + eBD[h[ca]])).sum())
    lg = (x[ch] * (np.log(lh32[ch].astype(np.float64)).astype(f32) - f32(np.log(15.0)))).sum() - (x[ch] * (eh[ch] - np.log(15.0))).sum()
    print(f"  table correction booked for clipped lanes (spurious)              : {sp:+.3e}  (x5 with the device's exp2-based tables)")
    print(f"  k log(raw rate) in float32 at clipped rates (home side)           : {lg:+.3e}")
