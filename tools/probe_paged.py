"""Single-pass paged partition (csrc/ek_paged.h: k_page_partition + k_page_directory) -- correctness of the page lists on small
inputs of every kind (masks, 64-bit indices, skew, ragged sizes, 2 .. 256 buckets), then time at the headline size against the
count / scan / partition pipeline of the product.  GPU box: python tools/probe_paged.py [check|time|all]"""
import ctypes, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream(); P = ctypes.c_void_p
pl = capi.probe_lib()
mode = sys.argv[1] if len(sys.argv) > 1 else "all"


CB = 8 * 2 * 256 + 256          # words of the counter block (csrc/ek_paged.h: kPgCounterWords)


def pcheck(rc):
    if rc != 0:
        raise RuntimeError(pl.ek_hip_probe_last_error().decode())


class Run:
    def __init__(self, idx_h, x_h, K, shift, mask_h=None, target_pieces=256):
        self.n, self.K, self.shift = len(idx_h), K, shift
        geo = (ctypes.c_uint64 * 8)()
        pcheck(pl.ek_hip_probe_page_plan(ctypes.c_size_t(self.n), ctypes.c_size_t(K), shift, geo))
        self.ps, self.cap, self.W, self.slots, self.chunk, self.page_slots, self.lds, self.nb = [int(v) for v in geo]
        self.idx64 = idx_h.dtype.itemsize == 8
        self.idx, self.x = capi.Buf.from_numpy(idx_h), capi.Buf.from_numpy(x_h)
        self.mask = capi.Buf.from_numpy(mask_h.astype(np.uint8)) if mask_h is not None else None
        page = 1 << self.ps
        self.lp = capi.Buf(np.uint16, self.page_slots * page)
        self.xp = capi.Buf(np.float32, self.page_slots * page)
        self.wdir = capi.Buf(np.uint32, self.page_slots)
        self.wlist = capi.Buf(np.uint32, self.page_slots)
        self.gfull = capi.Buf(np.uint32, self.page_slots)
        self.gpart = capi.Buf(np.uint32, self.W * self.nb + 1)
        self.meta = capi.Buf(np.uint32, CB + 3 * 257 + 3 * self.nb * self.W)
        self.tp = target_pieces
        self.dbg = capi.Buf(np.uint64, self.W * 24)

    def launch(self, nts=0, directory=1):
        pcheck(pl.ek_hip_probe_page_partition(nts, int(self.idx64), P(self.idx.ptr), P(self.x.ptr), P(self.mask.ptr if self.mask else None),
                                              ctypes.c_size_t(self.n), ctypes.c_size_t(self.K), self.shift, ctypes.c_uint32(self.tp),
                                              P(self.lp.ptr), P(self.xp.ptr), P(self.wdir.ptr), P(self.wlist.ptr), P(self.gfull.ptr),
                                              P(self.gpart.ptr), P(self.meta.ptr), directory, P(self.dbg.ptr)))

    def verify(self, idx_h, x_h, mask_h=None):
        capi.sync()
        page = 1 << self.ps
        lp, xp = self.lp.numpy(), self.xp.numpy()
        gfull, gpart, meta = self.gfull.numpy(), self.gpart.numpy(), self.meta.numpy()
        bf, bp, pp = meta[CB:CB + 257], meta[CB + 257:CB + 514], meta[CB + 514:CB + 771]
        nb = self.nb
        on = np.ones(self.n, bool) if mask_h is None else mask_h.astype(bool)
        keys_in = idx_h[on].astype(np.uint64)
        exp = np.sort(keys_in * np.uint64(1 << 32) + x_h[on].view(np.uint32).astype(np.uint64))
        got = []
        seen = np.zeros(self.page_slots, np.int32)
        for b in range(nb):
            f = gfull[bf[b]:bf[b + 1]].astype(np.int64)
            np.add.at(seen, f, 1)
            pos = (f[:, None] * page + np.arange(page)[None, :]).ravel()
            k = (np.uint64(b) << np.uint64(self.shift)) + lp[pos].astype(np.uint64)
            got.append(k * np.uint64(1 << 32) + xp[pos].view(np.uint32).astype(np.uint64))
            for e in gpart[bp[b]:bp[b + 1]]:
                pg, cntm1 = int(e) >> 6, int(e) & 63
                seen[pg] += 1
                pos = pg * page + np.arange(cntm1 + 1)
                k = (np.uint64(b) << np.uint64(self.shift)) + lp[pos].astype(np.uint64)
                got.append(k * np.uint64(1 << 32) + xp[pos].view(np.uint32).astype(np.uint64))
        got = np.sort(np.concatenate(got)) if got else np.zeros(0, np.uint64)
        ok = len(got) == len(exp) and np.array_equal(got, exp) and seen.max(initial=0) <= 1
        pieces_ok = pp[0] == 0 and np.all(np.diff(pp[:nb + 1].astype(np.int64)) >= 0)
        return bool(ok and pieces_ok), int(bf[nb]), int(bp[nb]), int(pp[nb])


rng = np.random.default_rng(0)
if mode in ("check", "all"):
    print("# correctness: multiset of (key, x) over the page lists == active input; no page listed twice")
    cases = []
    for name, n, K, shift in (("uniform 128 buckets", 1 << 21, 1 << 20, 13), ("ragged n", (1 << 20) + 12345, 1 << 20, 13),
                              ("256 buckets (32-element pages)", 1 << 21, 1 << 21, 13), ("2 buckets", 1 << 20, 1 << 15, 14),
                              ("64 buckets", 1 << 21, 1 << 20, 14), ("tiny", 1000, 1 << 20, 13), ("K ragged, 77 buckets", 1 << 20, 77 * 8192 - 5, 13)):
        idx_h = rng.integers(0, K, n).astype(np.uint32)
        cases.append((name, idx_h, K, shift, None))
    n, K = 1 << 21, 1 << 20
    cases.append(("75 % mask", rng.integers(0, K, n).astype(np.uint32), K, 13, rng.integers(0, 4, n) != 0))
    cases.append(("64-bit indices", rng.integers(0, K, n).astype(np.uint64), K, 13, None))
    cases.append(("one index", np.full(n, 12345, np.uint32), K, 13, None))
    z = np.minimum(rng.zipf(1.3, n), K) - 1
    cases.append(("zipf(1.3)", z.astype(np.uint32), K, 13, None))
    cases.append(("sorted indices", np.sort(rng.integers(0, K, n)).astype(np.uint32), K, 13, None))
    for name, idx_h, K, shift, mask_h in cases:
        x_h = rng.uniform(-1, 1, len(idx_h)).astype(np.float32)
        for nts in (0,):
            r = Run(idx_h, x_h, K, shift, mask_h)
            r.launch(nts)
            ok, nf, npart, npieces = r.verify(idx_h, x_h, mask_h)
            print(f"{name:32s} nts={nts}  W={r.W:4d} cap={r.cap:5d} page={1 << r.ps:3d}  full {nf:7d} partial {npart:6d} pieces {npieces:4d}  {'ok' if ok else 'MISMATCH'}")
            # run to run: the page lists do not depend on timing
            def content():
                g = r.gfull.numpy()[:nf].astype(np.int64)
                pos = (g[:, None] * (1 << r.ps) + np.arange(1 << r.ps)[None, :]).ravel()
                return r.lp.numpy()[pos], r.xp.numpy()[pos]
            a = content()
            r.launch(nts); capi.sync()
            b2 = content()
            if not (np.array_equal(a[0], b2[0]) and np.array_equal(a[1], b2[1])):
                print("    the elements in list order differ between two runs")

if mode in ("time", "all"):
    n, K = int(os.environ.get("PROBE_N", 1 << 26)), 1 << 20
    idx_h = rng.integers(0, K, n).astype(np.uint32)
    x_h = rng.uniform(-1, 1, n).astype(np.float32)
    print(f"# timing: {n >> 20} Mi elements, K = {K >> 20} Mi, 128 buckets of 8 Ki; 14 B/elt")
    shift_t = int(os.environ.get("PROBE_SHIFT", 13))          # 13: 128 buckets of 8 Ki entries (64-element pages); 12: 256 buckets of 4 Ki (32-element pages)
    r = Run(idx_h, x_h, K, shift_t)
    print(f"# shift {shift_t}: {r.nb} buckets, pages of {1 << r.ps} elements, cap {r.cap}, {r.slots} page slots per workgroup")
    for nts in (0, 0):
        for d in (0, 1):
            f = lambda: r.launch(nts, d)
            ms = statistics.median(hiprt.time_region(st, f, iters=10, warmup=2) for _ in range(5))
            print(f"paged partition nts={nts} directory={d}: {ms:7.4f} ms  {n * 14 / ms / 1e9:6.3f} TB/s")
    ok, nf, npart, npieces = r.verify(idx_h, x_h)
    print(f"verify at {n >> 20} Mi:", "ok" if ok else "MISMATCH", nf, npart, npieces)
    if os.environ.get("EK_PG_TIMING"):                 # probe library built with -DEK_PG_TIMING
        r.launch(0, 0); capi.sync()
        draw = r.dbg.numpy()
        d = draw[:r.W * 16].reshape(r.W, 2, 8).astype(np.float64)
        ts = draw[r.W * 16:r.W * 20].reshape(r.W, 4).astype(np.int64)
        te = draw[r.W * 20:].reshape(r.W, 4).astype(np.int64)
        print(f"  epilogue (us, median): bookkeeping {np.median(te[:,0]-ts[:,2])/100:.1f}, wait stores + barrier {np.median(te[:,1]-te[:,0])/100:.1f}, partial pages {np.median(te[:,2]-te[:,1])/100:.1f}, lists {np.median(ts[:,3]-te[:,2])/100:.1f}")
        t0 = ts[:, 0].min()
        us = lambda v: (v - t0) / 100.0
        print(f"  wall clock (100 MHz), us from the first start: starts {us(ts[:,0]).min():.1f}..{us(ts[:,0]).max():.1f}, loop begins {us(ts[:,1]).min():.1f}..{us(ts[:,1]).max():.1f}, loop ends {us(ts[:,2]).min():.1f}..{np.median(us(ts[:,2])):.1f}..{us(ts[:,2]).max():.1f}, ends {us(ts[:,3]).min():.1f}..{np.median(us(ts[:,3])):.1f}..{us(ts[:,3]).max():.1f}")
        ends = us(ts[:, 2])
        print("  loop end by XCD (block % 8), us: " + ", ".join(f"{x}: {np.median(ends[x::8]):.1f} (max {ends[x::8].max():.1f})" for x in range(8)))
        order = np.argsort(ends)
        print("  slowest blocks:", [(int(b), round(float(ends[b]), 1)) for b in order[-12:]])
        names = ["wait loads", "placement", "barrier 1", "overflow", "-", "write-out", "barrier 2", "-"]
        tiles = r.chunk // 4096
        for wv in (0, 1):
            print(f"  wave {wv}: core cycles per tile (mean over workgroups) " + ", ".join(f"{names[k]} {d[:, wv, k].mean() / tiles:7.1f}" for k in range(7)) + f"  total {d[:, wv, :7].sum(axis=1).mean() / tiles:8.1f}")
    for name, gen in (("zipf(1.3)", lambda: (np.minimum(rng.zipf(1.3, n), K) - 1).astype(np.uint32)), ("one index", lambda: np.full(n, 777, np.uint32))):
        if os.environ.get("PROBE_SKIP_SKEW"): break
        r2 = Run(gen(), x_h, K, 13)
        f = lambda: r2.launch(0, 1)
        ms = statistics.median(hiprt.time_region(st, f, iters=5, warmup=1) for _ in range(3))
        print(f"paged partition, {name}: {ms:7.4f} ms")
        del r2
    # the product's pipeline on the same input
    A = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
    B = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
    def product():
        b = capi.Bucketed("fmadd", A, r.x, B, r.idx, hints=capi.Bucketed.HINT_ADJOINT)
        b.destroy()
    for _ in range(3): product()
    capi.sync()
    capi.profile_begin()
    for _ in range(5): product()
    tot = 0
    for k in sorted(capi.profile_end(), key=lambda k: -k["total_ms"]):
        if k["launches"]:
            print(f"    product {k['kernel']:26s} {k['total_ms'] / k['launches']:8.4f} ms")
            tot += k["total_ms"] / 5
    print(f"    product total {tot:8.4f} ms")
