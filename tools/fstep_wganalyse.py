"""Per-workgroup timing of the chunked launch (DSGD_PLAN_PROF=1 DSGD_FSTEP_DUMP=<file> python tools/fstep_wgdump.py <rows>):
start spread, durations by XCD, what the launch waits for."""
import sys
import numpy as np

runs, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur:
            runs.append(cur)
        cur = []
        continue
    cur.append([int(x) for x in line.split()])
if cur:
    runs.append(cur)
MHZ = float(sys.argv[2]) if len(sys.argv) > 2 else 2350.0   # readcyclecounter: the shader clock, one counter PER XCD (not synchronised)
for r in runs[-3:]:
    wg = np.asarray([q[0] for q in r], dtype=np.float64)
    st = np.asarray([q[1] for q in r], dtype=np.float64)
    en = np.asarray([q[2] for q in r], dtype=np.float64)
    hot = np.asarray([q[3] for q in r], dtype=np.float64)
    xcc = np.asarray([q[4] & 0xff for q in r], dtype=np.float64)
    tiles = np.asarray([(q[4] >> 8) & 0xffff for q in r], dtype=np.float64)
    ctiles = np.asarray([(q[4] >> 24) & 0xffff for q in r], dtype=np.float64)
    rows = np.asarray([(q[4] >> 40) & 0xffff for q in r], dtype=np.float64)
    longs = np.asarray([q[4] >> 56 for q in r], dtype=np.float64)
    t0 = st.min()                           # start / end: wall_clock64 (100 MHz, device-wide); hot tiles: shader cycles
    st, en = (st - t0) / 100.0, (en - t0) / 100.0
    dur = en - st
    print("workgroups %d: start spread %.1f us (p50 %.1f, p90 %.1f), duration avg %.1f p50 %.1f p90 %.1f max %.1f, last end %.1f us; hot tiles avg %.1f max %.1f" % (
        len(wg), st.max(), np.median(st), np.quantile(st, 0.9), dur.mean(), np.median(dur), np.quantile(dur, 0.9), dur.max(), en.max(), (hot / MHZ).mean(), (hot / MHZ).max()))
    for x in sorted(set(xcc.astype(int))):
        m = xcc == x
        print("   xcc %d: %3d wgs, start avg %.1f, duration avg %.1f max %.1f, end max %.1f" % (x, m.sum(), st[m].mean(), dur[m].mean(), dur[m].max(), en[m].max()))
    dur0 = en - st
    A = np.stack([np.ones_like(tiles), tiles, ctiles, rows, longs], axis=1)
    coef, *_ = np.linalg.lstsq(A, dur0, rcond=None)
    fit = A @ coef
    print("   chunk contents: hot tiles %.0f..%.0f (avg %.1f), cold tiles %.0f..%.0f, rows %.0f..%.0f, long rows max %.0f" % (
        tiles.min(), tiles.max(), tiles.mean(), ctiles.min(), ctiles.max(), rows.min(), rows.max(), longs.max()))
    print("   duration ~ %.2f + %.4f * hot tiles + %.4f * cold tiles + %.5f * rows + %.3f * long rows: residual std %.2f us (duration std %.2f); corr(duration, hot tiles) %.3f; ceil(hot tiles / 16) %.0f..%.0f" % (
        coef[0], coef[1], coef[2], coef[3], coef[4], (dur0 - fit).std(), dur0.std(), np.corrcoef(dur0, tiles)[0, 1], np.ceil(tiles / 16).min(), np.ceil(tiles / 16).max()))
    order = np.argsort(-en)[:8]
    print("   latest: " + ", ".join("wg %d (xcc %d) start %.1f dur %.1f hot %.1f" % (wg[i], xcc[i], st[i], dur[i], hot[i] / MHZ) for i in order))
