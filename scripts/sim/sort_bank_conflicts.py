"""Offline simulation of the LDS bank behaviour of the per-tile bucket sort (pegasus_amd/csrc/tilebin.hip.h
bucket_sort_tile) on the real lists of a full-size view, for the layouts round 5 considered (round-4 verdict item 3).

Per wave-instruction a wave64 LDS access is served in lane groups (MI355X_MICROARCH.md, LDS table): 4-byte accesses in two
groups of 32 lanes over 32 banks ((addr / 4) mod 32); 8-byte reads in two groups of 32 over 64 banks; 8-byte writes in
four groups of 16 over 32 banks.  A group costs max over banks of the number of DISTINCT addresses on that bank (equal
addresses broadcast for reads; for returning atomics equal addresses serialise, counted separately).

    python scripts/sim/sort_bank_conflicts.py [c3|c5] [view index]

The unsorted list of a tile is taken in ascending resident (Morton) index -- the order the scatter walk's chunks reserve
their slices in; inside a chunk the arrival order is not deterministic on the device, so this is the model, not a replay.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402  (offline analysis script: allowed like the other scripts/sim tools)
from pegasus_amd import scenes  # noqa: E402
from pegasus_amd.scene_order import spatial_order  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
vi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cloud, views = (scenes.scene_c5 if wl == "c5" else scenes.scene_c3)(n_views=max(vi + 1, 4))
act = cloud.activated()
perm = spatial_order(act["means3d"], cloud.object_id)
act = {k: np.ascontiguousarray(a[perm]) for k, a in act.items()}
v = views[vi]
o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1, tie_index=perm)
depth_bits = o["depth"].view(np.uint32)
ranges = o["ranges"].astype(np.int64)
gs = o["gauss_sorted"]

TIERS = [(256, 2, 512), (256, 4, 1024), (256, 8, 2048), (512, 8, 4096), (1024, 8, 8192), (1024, 16, 16000)]


def tier_of(n):
    for th, e, cap in TIERS:
        if n <= cap:
            return th, e, cap
    return None


def group_cost(addr_words, valid, lanes_per_group, banks, same_addr_serialises):
    """addr_words [W, 64] word addresses of one wave-instruction per row; returns (ideal cycles, actual cycles)."""
    W = addr_words.shape[0]
    ideal = actual = 0
    for g0 in range(0, 64, lanes_per_group):
        a = addr_words[:, g0:g0 + lanes_per_group]
        m = valid[:, g0:g0 + lanes_per_group]
        any_ = m.any(axis=1)
        ideal += int(any_.sum())
        bank = a % banks
        for w in np.nonzero(any_)[0]:
            aw, bw = a[w][m[w]], bank[w][m[w]]
            if same_addr_serialises:
                cnt = np.bincount(bw, minlength=banks)
            else:
                uniq = np.unique(aw)
                cnt = np.bincount(uniq % banks, minlength=banks)
            actual += int(cnt.max())
    return ideal, actual


def simulate(layouts):
    tot = {name: {"hist": [0, 0], "hist_same": [0, 0], "fin": [0, 0], "keyw": [0, 0], "rank": [0, 0], "idxw": [0, 0]} for name in layouts}
    n_lists = 0
    rng = np.random.default_rng(0)
    tiles = np.nonzero(ranges[:, 1] - ranges[:, 0] > 64)[0]
    if len(tiles) > 400:
        tiles = rng.choice(tiles, 400, replace=False)
    for t in tiles:
        idx = np.sort(gs[ranges[t, 0]:ranges[t, 1]])            # scatter order model: ascending resident index
        n = len(idx)
        tier = tier_of(n)
        if tier is None:
            continue
        TH, E, CAP = tier
        NB = 8192 if CAP == 16000 else CAP
        d = depth_bits[idx].astype(np.int64)
        mn, mx = d.min(), d.max()
        scale = np.float32(NB) / (np.float32(mx - mn) + np.float32(1.0))
        b = np.minimum(((d - mn).astype(np.float32) * scale).astype(np.int64), NB - 1)
        n_lists += 1
        # arrival slot inside the bucket (atomic order = list order in this model)
        order = np.argsort(b, kind="stable")
        counts = np.bincount(b, minlength=NB)
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        slot = np.empty(n, np.int64)
        slot[order] = np.arange(n) - starts[b[order]]
        pos = starts[b] + slot                                   # where the key is parked (s_keys index)
        key = (d << 32) | idx.astype(np.int64)
        final = np.empty(n, np.int64)
        final[np.argsort(key)] = np.arange(n)                    # sorted position (s_idx index)
        for name, (assign, swz) in layouts.items():
            # which list element lane l of wave w handles at step e
            pad = TH * E
            elem = np.full(pad, -1, np.int64)
            if assign == "striped":                              # element e * TH + t  (today)
                elem[:n] = np.arange(n)
                grid = elem.reshape(E, TH)                       # [e, t]
            else:                                                # blocked: thread t owns elements t * E + e
                elem[:n] = np.arange(n)
                grid = elem.reshape(TH, E).T
            rows = grid.reshape(E * (TH // 64), 64)              # one wave-instruction per row
            valid = rows >= 0
            r = np.where(valid, rows, 0)
            bb = b[r]
            if swz == "xor":                                     # bucket word address swizzle: rotate the bank by the row
                hb = bb ^ ((bb >> 5) & 31)
            else:
                hb = bb
            for kname, aw, lpg, banks, same in (("hist", hb, 32, 32, False), ("hist_same", hb, 32, 32, True),
                                                 ("fin", hb, 32, 32, False)):
                i, a = group_cost(aw, valid, lpg, banks, same)
                tot[name][kname][0] += i; tot[name][kname][1] += a
            i, a = group_cost(2 * pos[r], valid, 16, 32, False)  # ds_write_b64 of the parked key: 4 x 16 lanes, 32 banks, 2 words
            tot[name]["keyw"][0] += 2 * i; tot[name]["keyw"][1] += 2 * a
            multi = valid & (counts[bb] > 1)
            i, a = group_cost(2 * starts[bb], multi, 32, 64, False)   # first member read (ds_read_b64: 64 banks)
            tot[name]["rank"][0] += i; tot[name]["rank"][1] += a
            i, a = group_cost(final[r], valid, 32, 32, False)    # ds_write_b32 of the index image
            tot[name]["idxw"][0] += i; tot[name]["idxw"][1] += a
    return tot, n_lists


layouts = {"today (striped keys, plain buckets)": ("striped", None), "blocked keys": ("blocked", None),
           "striped keys, xor-swizzled buckets": ("striped", "xor"), "blocked keys, xor-swizzled buckets": ("blocked", "xor")}
tot, n_lists = simulate(layouts)
print(f"# {wl} view {vi}: {int(o['num_instances'])} instances, {n_lists} sampled lists > 64 keys; LDS-array cycles per phase, "
      f"conflict share = (actual - ideal) / actual")
for name, ph in tot.items():
    print(name)
    a_sum = i_sum = 0
    for k, (i, a) in ph.items():
        if k == "hist_same":
            print(f"    {k:10s} ideal {i:9d} actual {a:9d}  x{a / max(i, 1):.2f}   (if equal addresses of an atomic serialise)")
            continue
        a_sum += a; i_sum += i
        print(f"    {k:10s} ideal {i:9d} actual {a:9d}  x{a / max(i, 1):.2f}")
    print(f"    {'sum':10s} ideal {i_sum:9d} actual {a_sum:9d}  conflict share {(a_sum - i_sum) / a_sum:.3f}")
