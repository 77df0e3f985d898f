#!/usr/bin/env python
"""bench.py -- the north-star measurement: Gelem/s and %HBM-roofline of a 64 Mi-element DiffArray
forward + backward() on MI355X (BASELINE.json: metric / configs[2]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3b|cfg3a|cfg2] [--n 67108864]

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:

  cfg3b (default, BASELINE configs[2] "with scatter_add grads"):
        A, B leaves of size K = 1 Mi (replicated); idx = hash mod K; x plain, size N
        a = gather(A, idx); b = gather(B, idx); y = hsum(sin(fmadd(a, x, b))); backward(y)
        -> grad_A, grad_B (K) via scatter_add
  cfg3a a, b leaves of size N: y = hsum(sin(fmadd(a, x, b))); backward(y)
  cfg2  plain HIPArray: y = hsum(sin(exp(fmadd(a, x, b))))

`value` is the headline workload (cfg3b unless --workload says otherwise).  On one GPU the other two GPU
configurations of BASELINE.md section 4 are measured in the same run and reported under "also" (same timing
method), because the three have separate pass lines.

Multi-GPU (one process per GPU; `python bench.py --gpus N` starts its own ranks): the arrays are index-range
sharded, the K-element tables are replicated; per step ONE asynchronous RCCL all-reduce finishes y and the table
gradients (enoki_amd/dist.py).  Default --scaling strong -- what BASELINE.json's metric, `north_star` and BASELINE.md
section 4 state: --n (64 Mi) elements IN TOTAL, --n / N per GPU; `value` = --n / max-over-ranks time.  The same run then
also measures the weak form (every GPU owns --n elements of an array of N x --n) and carries it as the labelled
sub-record `weak` of the same line (--no-weak skips it; --scaling weak makes it the `value` as in round 5).

The JSON line carries, besides the contract fields, `roofline` for the dominant kernel (live per-launch timing
with HIP events on the library stream, ek_hip_profile_*) and `cpu_baseline` (the reference build oracle/_ref, or
the C restatement when it is absent, timed on ONE host core -- the reference has no threading).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_TBS = 8.0           # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
K_TABLE = 1 << 20
DESCRIPTION = {
    "cfg3b": "DiffArray<HIPArray<float>> y=hsum(sin(a*x+b)), a=gather(A,idx), b=gather(B,idx), K=1Mi; backward() with scatter_add grads",
    "cfg3a": "DiffArray<HIPArray<float>> y=hsum(sin(a*x+b)); backward(), a,b leaves of size N",
    "cfg2": "HIPArray<float> hsum(sin(exp(fmadd(a,x,b)))): ONE pass over a, x, b (the fma and both maps stay unevaluated until the reduction consumes the chain, ek_hip_reduce_chain)",
    "cfg4": "ray-sphere (tests/sphere.cpp:58-83): masked gather of a 16384^2-style pixel grid through a random "
            "permutation, make_rays/intersect_rays/shade_hits, masked scatter, count(hit); 32 Mi rays per GPU; ONE fused "
            "kernel through enoki::vectorize() (examples/sphere_fused.cpp)",
    "cfg4_packed": "cfg4's fused kernel over a pixel grid stored as packed {x, y} records: ONE 8-byte lookup per ray "
                   "instead of two 4-byte ones (examples/sphere_fused.cpp sphere_fused_packed_device), bit-identical image",
    "cfg4_unfused": "the same program on Array<HIPArray<float>,3> op by op (~40 eager kernels), bit-identical image",
    "cfg4_bucketed": "cfg4 executed per PIXEL instead of per ray (enoki::vectorize_through, examples/sphere_fused.cpp "
                     "sphere_through_device): gather and scatter share the permutation and the kernels depend on the gathered pixel only, "
                     "so the rays are partitioned by pixel bucket once and every bucket streams its grid slice in and its image slice "
                     "out in order; bit-identical image and hit count",
}
# the neighbours of the headline step (same chain, one thing changed): what a caller meets one step off the benchmark's exact
# expression.  Parity of each against the reference build at 64 Mi elements: tests/test_headline_parity_gpu.py.
CFG3B_VARIANTS = {
    "cfg3b_cos": dict(func="cos"), "cfg3b_exp": dict(func="exp"), "cfg3b_seed3": dict(seed=3.0), "cfg3b_masked": dict(masked=True),
    "cfg3b_i64": dict(idx64=True), "cfg3b_K2Mi": dict(K=1 << 21), "cfg3b_K4Mi": dict(K=1 << 22), "cfg3b_K16Mi": dict(K=1 << 24),
    "cfg3b_sqrt": dict(func="sqrt", shift=3.0),          # (B + 3: u > 0; the derivative's factor .5 / sqrt(u) is a function of u)
    "cfg3b_rcp": dict(func="rcp", shift=3.0),            # (the derivative's factor -sqr(rcp(u)) as ONE map of u)
    # u written with OPERATORS, the literal spelling of BASELINE.json configs[2] (`y=hsum(sin(a*x+b))`): a product and a sum with a
    # rounding each -- the same bucket-ordered step (round 4: 48 Gelem/s in element order)
    "cfg3b_operators": dict(spelling="a*x+b"),
    "cfg3b_operators_sub": dict(spelling="b-a*x", func="cos"),
    # ONE gather times an array (`texture lookup * weight`, the shape of cfg5's albedo lookups): y = hsum(sin(gather(A, idx) * x))
    "cfg3b_product": dict(spelling="a*x"),
    # SKEWED indices (a textured scene has hot texels): log-uniform over K = 1 Mi -- a magnitude m uniform in 0..19, then an index
    # uniform in [2^m - 1, 2^(m + 1) - 1): density ~ 1 / k like Zipf(s = 1), integer arithmetic only (host and device generate the same
    # bits).  Half of the lookups fall into the first 1023 entries, 5 % of them on entry 0.
    "cfg3b_zipf": dict(zipf=True),
}
for _w, _v in CFG3B_VARIANTS.items():
    DESCRIPTION[_w] = ("cfg3b with " + ", ".join(f"{k}={v}" for k, v in _v.items()) +
                       " (y = seed * hsum(func(fmadd(gather(A, idx, mask), x, gather(B, idx, mask)))), backward(); 75 % mask of SURVEY 8d; spelling: u written with operators)")
N_RAYS_PER_GPU = 1 << 25
N_PATHS_PER_GPU = 1 << 24
DESCRIPTION["cfg5_unfused"] = "cfg5 spelled op by op with the python bindings (every operation one kernel, three gather nodes on the tape)"
DESCRIPTION["cfg5_cpp"] = "cfg5: the template of examples/path_trace.h on DiffArray<HIPArray<float>>, op by op"
DESCRIPTION["cfg5"] = ("synthetic 3-bounce path tracer inside a textured unit sphere (SURVEY 8d cfg5, not in the reference): "
                       "PCG32 sampling, sphere intersection, (theta, phi) -> texel, differentiable gather of the albedo "
                       "texture (K = 1 Mi), cosine-weighted bounce; loss = hsum(radiance); backward() scatter_adds the "
                       "texture gradient; 16 Mi paths per GPU; ONE fused kernel per evaluation (examples/path_trace.cpp: the template "
                       "of examples/path_trace.h on one-element packets with forward-mode duals), one tape node, one scatter_add")
# kernel name reported by the library -> kernel symbol prefix in the rocprofv3 PMC summary (profiles/)
PMC_SYMBOL = {"bucket_accumulate": "k_bucket_accumulate", "bucket_partition": "k_page_partition", "bucket_directory": "k_page_directory", "bucket_pair_fma_reduce": "k_bucket_pair_forward<",
              "bucket_pair_fma_reduce_adjoint": "k_bucket_pair_forward_adjoint",
              "bucket_count": "k_bin_count", "gather_pair_fmadd": "k_map_gathered<GTernary<0", "gather": "k_gather", "scatter_add_partition": "k_bin_partition", "scatter_add_accumulate": "k_bin_accumulate",
              "scatter_add_count": "k_bin_count", "fmadd": "k_map3<TernaryOp<0", "sincos": "k_map1x2<SinCosOp",
              "safe_mul": "k_map2<BinaryOp<13", "hsum": "k_reduce_stage1", "sin": "k_map1<UnaryOp<10", "exp": "k_map1<UnaryOp<12",
              "reduce_chain": "k_chain_reduce", "map_chain": "k_chain_map"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of about one second (0.34 ms per headline step), long enough for an outside sampler to see a busy GPU
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3b", choices=["cfg3b", "cfg3a", "cfg2", "cfg4", "cfg4_packed", "cfg4_unfused", "cfg4_bucketed", "cfg5", "cfg5_unfused", "cfg5_cpp"] + list(CFG3B_VARIANTS))
    ap.add_argument("--n", type=int, default=1 << 26, help="elements IN TOTAL (--scaling strong, the default: what BASELINE.md section 4 and north_star state) or per GPU (--scaling weak)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1 GPUs: skip the weak-scaling sub-record (`weak` in the line)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="cfg2 / cfg3a / cfg3b on N > 1 GPUs.  strong (default, the contract's configuration): --n elements in total, "
                         "--n / N per GPU; weak: every GPU owns an index range of --n elements of an array of N x --n "
                         "(DESIGN.md section 7 for what each can give).  With strong, the weak measurement rides along as `weak`.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads on one GPU")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--deterministic", action="store_true", help="bit-reproducible fp scatter_add (sorted path)")
    ap.add_argument("--eager", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--dump-gradients", default=None, metavar="DIR", help="cfg3b: every rank saves what it holds of the table gradients after the exchange (owned slices when reduce-scattered) as DIR/grad_rank<r>.npz -- for the multi-rank parity tests")
    ap.add_argument("--graph", action="store_true", help="ALSO time the K steps as replays of a captured step graph (graph_ms_per_step); `value` stays the python-driven protocol")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="(default since round 5: what BASELINE.json's north_star names) multi-GPU: all-reduce the table gradients, every rank ends up with all K bins")
    ap.add_argument("--reduce-scatter-grads", action="store_true",
                    help="multi-GPU: reduce-scatter the table gradients instead (rank r receives bins [r K / P, (r + 1) K / P): half the bytes, "
                         "for callers that update the slice they own)")
    ap.add_argument("--pre-warm-s", type=float, default=0.6,
                    help="seconds of the headline step run BEFORE the --warmup steps (clocks and caches in the state of a long run; "
                         "reported as pre_warm_s, outside the timed region)")
    # ranks started by relaunch_as_ranks() receive the original command line through the environment: torch.distributed.run parses
    # its own options with prefix matching, and a script argument like `--n` is ambiguous to it
    forwarded = os.environ.get("ENOKI_BENCH_ARGV")
    return ap.parse_args(json.loads(forwarded)) if forwarded is not None else ap.parse_args()


PMC_FILE = os.path.join("profiles", "rocprof_pmc_r06.txt")
KSTATS_FILE = os.path.join("profiles", "rocprof_kernel_stats_r06.txt")


def kernels_sha16():
    """fingerprint of the device code (every source that goes into libenoki-hip.so): the PMC summary under profiles/ is
    stamped with it by tools/rocprof_summary.py, and a summary taken from other kernels is refused"""
    from enoki_amd import _build
    return _build.kernels_sha16()


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, provenance) from the committed rocprofv3 PMC summary (separate --pmc passes,
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/rocprof_summary.py).  The summary names the kernel sources it was
    measured on (`# kernels_sha16`); when they are not the sources of THIS tree the number is not reported."""
    path = os.path.join(ROOT, PMC_FILE)
    sym = PMC_SYMBOL.get(kernel)
    if not sym or not os.path.exists(path):
        return None, f"no PMC summary for kernel '{kernel}' ({PMC_FILE})"
    stamp, best = None, None
    for line in open(path):
        if line.startswith("# kernels_sha16:"):
            stamp = line.split(":", 1)[1].strip()
        if line.startswith(sym):
            f = line.split()
            try:
                grid, rd, wr = int(f[-4]), float(f[-2]), float(f[-1])
            except ValueError:
                continue
            if best is None or grid > best[0]:
                best = (grid, (rd + wr) * 1e6)
    here = kernels_sha16()
    if stamp != here:
        return None, f"{PMC_FILE} was measured on kernel sources {stamp}, this tree is {here}: refused"
    if not best:
        return None, f"{PMC_FILE} has no line for {sym}"
    return int(best[1]), (f"{PMC_FILE} (kernels_sha16 {stamp}): separate rocprofv3 --pmc passes over the same command, "
                          "2 x FETCH_SIZE + WRITE_SIZE per launch")


def rocprof_avg_us(kernel):
    """average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary (no launch gaps), when that
    summary was taken on THIS tree's kernel sources"""
    path = os.path.join(ROOT, KSTATS_FILE)
    sym = PMC_SYMBOL.get(kernel)
    if not sym or not os.path.exists(path):
        return None, f"no kernel-trace summary for '{kernel}' ({KSTATS_FILE})"
    stamp, best = None, None
    for line in open(path):
        if line.startswith("# kernels_sha16:"):
            stamp = line.split(":", 1)[1].strip()
        if line.startswith(sym):
            f = line.split()
            try:
                calls, total, avg = int(f[-4]), float(f[-3]), float(f[-2])
            except ValueError:
                continue
            if best is None or total > best[0]:
                best = (total, avg)
    if stamp != kernels_sha16():
        return None, f"{KSTATS_FILE} was measured on kernel sources {stamp}, this tree is {kernels_sha16()}: refused"
    if not best:
        return None, f"{KSTATS_FILE} has no line for {sym}"
    return best[1], f"{KSTATS_FILE} (kernels_sha16 {stamp})"


def rocprof_step_sum_us():
    """sum over the kernels of one headline step of their average duration in the committed rocprofv3 kernel trace (same stamp
    rule as rocprof_avg_us); every library kernel of the step is launched once per step"""
    path = os.path.join(ROOT, KSTATS_FILE)
    if not os.path.exists(path):
        return None
    stamp, total = None, 0.0
    for line in open(path):
        if line.startswith("# kernels_sha16:"):
            stamp = line.split(":", 1)[1].strip()
        if line.startswith("# step_sum_us:"):
            total = float(line.split(":", 1)[1])
    return total if stamp == kernels_sha16() and total > 0 else None


def trace_check(ms_per_step, live_sum_ms, headline, profiled_ms_per_step=None):
    """Does the per-kernel evidence add up to the claimed step?  sum(kernels) <= step <= 1.1 x sum, for the live event deltas of
    this run and -- on the headline at 64 Mi elements -- for the committed rocprofv3 trace (tools/profile_r06.sh takes it after
    the same pre-warm)."""
    # step_over_live_sum: the event deltas against the wall time of the very steps they were taken in (recording an event per launch
    # slows those steps a little: against the unprofiled step the sum used to come out LARGER than what it decomposes);
    # timed_step_over_live_sum: against the timed (unprofiled) step
    prof_ms = profiled_ms_per_step if profiled_ms_per_step else ms_per_step
    rep = {"ms_per_step": round(ms_per_step, 4), "profiled_ms_per_step": round(prof_ms, 4), "sum_live_kernel_ms": round(live_sum_ms, 4),
           "step_over_live_sum": round(prof_ms / live_sum_ms, 3) if live_sum_ms > 0 else None,
           "timed_step_over_live_sum": round(ms_per_step / live_sum_ms, 3) if live_sum_ms > 0 else None}
    rp = rocprof_step_sum_us() if headline else None
    if rp:
        rep["sum_rocprof_kernel_ms"] = round(rp * 1e-3, 4)
        rep["step_over_rocprof_sum"] = round(ms_per_step / (rp * 1e-3), 3)
        rep["ok"] = bool(0.97 * rp * 1e-3 <= ms_per_step <= 1.1 * rp * 1e-3)
        if not rep["ok"]:
            print(f"[bench] trace check: step {ms_per_step:.4f} ms against {rp * 1e-3:.4f} ms of traced kernels ({KSTATS_FILE})", file=sys.stderr)
    return rep


def path_trace(ek, ekc, tex, n, seed, first_lane=0, bounces=3, width=1024, record=None):
    """cfg5 op by op: examples/path_trace.h (the ONE source of the program: the reference-side oracle `ref_cfg5` instantiates that
    template on the reference's arrays, the fused kernel on one-element packets) spelled with the python bindings, operation for
    operation in the same order -- only parity class A operations decide where a path goes (no normalize / rcp / rsqrt), so all
    three visit the same texels.  Geometry and sampling use plain arrays (enoki.hip); only the texture lookups are on the tape, so
    backward() is one scatter_add per bounce into grad(tex)."""
    import math
    F, U32, U64, V3 = ekc.Float32, ekc.UInt32, ekc.UInt64, ekc.Vector3f
    pi = float(np.float32(math.pi))
    f32 = lambda v: float(np.float32(v))
    def unit(v):                      # component by component: exact divisions (vector / value goes through rcp in the reference)
        l = ekc.sqrt(ekc.dot(v, v))
        return V3(v.x / l, v.y / l, v.z / l)
    rng = ekc.PCG32(U64(seed), U64.arange(n) + U64(first_lane))
    z = F(1.0) - F(2.0) * rng.next_float32()
    r = ekc.sqrt(ekc.max(F(0.0), F(1.0) - z * z))
    phi = F(f32(2.0 * np.float32(pi))) * rng.next_float32()
    s_, c_ = ekc.sincos(phi)
    d = V3(r * c_, r * s_, z)
    o = V3(F(0.1), F(0.2), F(-0.1))
    throughput = ek.Float32(1.0)
    radiance = ek.Float32(0.0)
    for _ in range(bounces):
        b = ekc.dot(o, d)
        c = ekc.dot(o, o) - F(1.0)
        t = ekc.sqrt(ekc.max(F(0.0), b * b - c)) - b                       # far root: we are inside the sphere
        p = unit(o + d * t)
        theta = ekc.acos(ekc.min(ekc.max(p.z, F(-1.0)), F(1.0)))
        ph = ekc.atan2(p.y, p.x)
        uu = ekc.fmadd(ph, F(f32(np.float32(0.5) / np.float32(pi))), F(0.5)); vv = theta * F(f32(np.float32(1.0) / np.float32(pi)))
        ix = ekc.min(U32(uu * F(float(width))), U32(width - 1)); iy = ekc.min(U32(vv * F(float(width))), U32(width - 1))
        texel = iy * U32(width) + ix
        if record is not None:
            record.append(texel)
        albedo = ek.gather(tex, ek.UInt32(texel))                          # the only differentiable operation
        radiance = radiance + throughput * albedo * ek.Float32(0.1)        # the surface emits a little of its colour
        throughput = throughput * albedo
        # cosine-weighted bounce around the inward normal nrm = -p (concentric disk mapping)
        nrm = p * F(-1.0)
        r1 = F(2.0) * rng.next_float32() - F(1.0); r2 = F(2.0) * rng.next_float32() - F(1.0)
        swap = ekc.abs(r1) < ekc.abs(r2)
        rad = ekc.select(swap, r2, r1)
        ratio = ekc.select(swap, r1, r2) / ekc.select(rad == F(0.0), F(1.0), rad)
        half_pi, quarter_pi = f32(np.float32(0.5) * np.float32(pi)), f32(np.float32(0.25) * np.float32(pi))
        ang = ekc.select(swap, F(half_pi) - F(quarter_pi) * ratio, F(quarter_pi) * ratio)
        sn, cs = ekc.sincos(ang)
        dx = rad * cs; dy = rad * sn
        dz = ekc.sqrt(ekc.max(F(0.0), F(1.0) - dx * dx - dy * dy))
        # orthonormal frame around nrm (Duff et al. 2017)
        sign = ekc.copysign(F(1.0), nrm.z)
        a = F(-1.0) / (sign + nrm.z)
        bb = nrm.x * nrm.y * a
        sx = V3(F(1.0) + sign * nrm.x * nrm.x * a, sign * bb, F(-1.0) * sign * nrm.x)
        ty = V3(bb, sign + nrm.y * nrm.y * a, F(-1.0) * nrm.y)
        d = unit(sx * dx + ty * dy + nrm * dz)
        o = p + nrm * F(1e-3)
    radiance = radiance + throughput                                        # leftover energy reaches a white environment
    return ek.hsum(radiance)


class Bench:
    def __init__(self, args, scaling=None):
        import torch
        import enoki_amd.hip as ekc
        import enoki_amd.hip_autodiff as ek
        from enoki_amd import dist as ekd, synth
        self.torch, self.ekc, self.ek, self.ekd, self.synth, self.args = torch, ekc, ek, ekd, synth, args
        self.rank, self.local_rank, self.world = ekd.init()
        if self.world != args.gpus:
            # never print a line whose n_gpus is not what was asked for (main() re-launches itself as --gpus ranks when it is
            # started without a launcher, so this is a launcher that was given another --nproc-per-node)
            raise SystemExit(f"[bench] --gpus {args.gpus} but the process group has WORLD_SIZE={self.world}: refusing to measure")
        torch.cuda.set_device(self.local_rank)          # torch's HIP runtime initialises first
        ek.hip_init(self.local_rank)
        ekd.adopt_torch_stream(ek)                        # kernels + RCCL ordered by one stream
        if args.deterministic:
            ek.hip_set_tuning("deterministic", 1)
        self.dev = torch.device("cuda", self.local_rank)
        scaling = scaling or args.scaling
        self.weak = scaling == "weak" or args.workload.startswith(("cfg4", "cfg5"))
        self.N = args.n * self.world if scaling == "weak" else args.n       # (cfg4 / cfg5 size their own per-GPU work)
        self.begin, self.end = ekd.shard_range(self.N, self.rank, self.world)
        self.n = self.end - self.begin
        self.sh = ekd.Sharded(ek, self.N, device=self.dev)     # horizontal results of the shards -> ONE all-reduce per step

    def group_world(self):
        """ranks the process group actually has (what the collectives run over), not what the command line says"""
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_initialized() else 1

    def backend(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            return "none (single process)"
        be = dist.get_backend()
        return "nccl (RCCL over xGMI)" if be == "nccl" else f"{be} ({self.world} ranks on {self.torch.cuda.device_count()} visible GPU(s))"

    def make_step(self, workload):
        """returns (step_fn, packer, out_dict); inputs are generated on the device (seeds of SURVEY.md 8d)"""
        ek, ekc, ekd, synth = self.ek, self.ekc, self.ekd, self.synth
        n, begin = self.n, self.begin
        x = synth.uniform_pm1(begin, n, 2)
        out = {}
        if workload == "cfg3b" or workload in CFG3B_VARIANTS:
            var = CFG3B_VARIANTS.get(workload, {})
            kt = var.get("K", K_TABLE)
            A0 = synth.uniform_pm1(0, kt, 6); B0 = synth.uniform_pm1(0, kt, 7)
            if var.get("shift"):
                B0 = B0 + ekc.Float32(float(var["shift"]))
            idx = ek.UInt32(synth.index_mod(begin, n, 4, kt))
            if var.get("zipf"):
                m = synth.hash_u32(begin, n, 8) % ekc.UInt32(20)
                lo = (ekc.UInt32(1) << m) - ekc.UInt32(1)
                idx = ek.UInt32(lo + (synth.hash_u32(begin, n, 9) & lo))
            if var.get("idx64"):
                idx = ek.UInt64(idx)
            mask = ek.Mask((synth.hash_u32(begin, n, 5) & ekc.UInt32(3)) != ekc.UInt32(0)) if var.get("masked") else None
            func, seed = getattr(ek, var.get("func", "sin")), var.get("seed", 1.0)
            xd = ek.Float32(x)
            packer = self.sh if ekd.active() else None

            def compute():
                A = ek.Float32(A0); B = ek.Float32(B0)
                ek.set_requires_gradient(A); ek.set_requires_gradient(B)
                if mask is not None:
                    a = ek.gather(A, idx, mask); b = ek.gather(B, idx, mask)
                else:
                    a = ek.gather(A, idx); b = ek.gather(B, idx)
                sp = var.get("spelling", "fmadd")
                u = ek.fmadd(a, xd, b) if sp == "fmadd" else a * xd + b if sp == "a*x+b" else a * xd if sp == "a*x" else b - a * xd
                y = ek.hsum(func(u))
                if seed != 1.0:
                    y = y * seed
                ek.backward(y)
                out["y"] = ek.detach(y)
                out["gA"] = ek.gradient(A)
                out["gB"] = ek.gradient(B) if sp != "a*x" else out["gA"]        # (the product alone: B does not take part)

            def exchange(reuse=False):
                # library-level sharding (enoki_amd.dist.Sharded): the loss partial and both table gradients are finished by
                # ONE asynchronous all-reduce; with a replayed step graph the sources keep their addresses -> same plan
                if packer:
                    if reuse and out.get("plan") is not None:
                        out["plan"].run()
                    else:
                        # default: ONE all-reduce finishes the loss and both gradient tables on every rank (north_star: "hsum and
                        # grad accumulation finished by RCCL all-reduce").  --reduce-scatter-grads: rank r receives bins
                        # [r K / P, (r + 1) K / P) of both tables in one collective (half the bytes, nothing K-sized replicated
                        # after the exchange) and the loss rides in an extra column of it
                        scat = self.args.reduce_scatter_grads
                        out["reduced"] = [self.sh.reduce(out["y"]), self.sh.reduce(out["gA"], scattered=scat),
                                          self.sh.reduce(out["gB"], scattered=scat)]
                        out["plan"] = self.sh.flush()
        elif workload == "cfg3a":
            a0 = synth.uniform_pm1(begin, n, 1); b0 = synth.uniform_pm1(begin, n, 3)
            xd = ek.Float32(x)
            packer = self.sh if ekd.active() else None

            def compute():
                a = ek.Float32(a0); b = ek.Float32(b0)
                ek.set_requires_gradient(a); ek.set_requires_gradient(b)
                y = ek.hsum(ek.sin(ek.fmadd(a, xd, b)))
                ek.backward(y)
                out["ga"], out["gb"] = ek.gradient(a), ek.gradient(b)
                out["y"] = ek.detach(y)

            def exchange(reuse=False):
                if packer:
                    if reuse and out.get("plan") is not None:
                        out["plan"].run()
                    else:
                        out["reduced"] = [self.sh.reduce(out["y"])]
                        out["plan"] = self.sh.flush()
        elif workload in ("cfg5", "cfg5_cpp"):
            # the templated program of examples/path_trace.h through examples/libpath_trace.so: cfg5 = ONE fused kernel per
            # evaluation (one-element packets + forward-mode duals, one tape node, one scatter_add in backward()); cfg5_cpp = the
            # same template on DiffArray<HIPArray<float>>, every operation one kernel
            import ctypes
            n5 = N_PATHS_PER_GPU
            tex0 = ekc.fmadd(synth.uniform_pm1(0, K_TABLE, 8), ekc.Float32(0.3), ekc.Float32(0.5))    # albedo in [0.2, 0.8)
            packer = self.sh if ekd.active() else None
            lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "examples", "libpath_trace.so"))
            fn = lib.path_trace_fused_device if workload == "cfg5" else lib.path_trace_device
            loss_buf, grad_buf = ekc.Float32.empty(1), ekc.Float32.empty(K_TABLE)
            P = ctypes.c_void_p

            def step():
                rc = fn(P(tex0.data_ptr()), ctypes.c_size_t(K_TABLE), ctypes.c_size_t(n5), ctypes.c_uint64(0x853c49e6748fea9b + self.rank),
                        ctypes.c_uint64(self.rank * n5), 3, ctypes.c_uint32(1024), P(loss_buf.data_ptr()), P(grad_buf.data_ptr()))
                if rc != 0:
                    raise RuntimeError(f"path_trace: rc = {rc}")
                out["y"], out["grad"] = loss_buf, grad_buf
                if packer:
                    out["reduced"] = [self.sh.reduce(ek.Float32(loss_buf)), self.sh.reduce(ek.Float32(grad_buf))]
                    self.sh.flush()
        elif workload == "cfg5_unfused":
            n5 = N_PATHS_PER_GPU
            tex0 = ekc.fmadd(synth.uniform_pm1(0, K_TABLE, 8), ekc.Float32(0.3), ekc.Float32(0.5))    # albedo in [0.2, 0.8)
            packer = self.sh if ekd.active() else None

            def step():
                tex = ek.Float32(tex0)
                ek.set_requires_gradient(tex)
                loss = path_trace(ek, ekc, tex, n5, seed=0x853c49e6748fea9b + self.rank, first_lane=self.rank * n5)
                ek.backward(loss)
                g = ek.gradient(tex)
                out["y"] = ek.detach(loss)
                if packer:
                    out["reduced"] = [self.sh.reduce(out["y"]), self.sh.reduce(g)]
                    self.sh.flush()
                out["grad"] = g
        elif workload in ("cfg4", "cfg4_packed", "cfg4_unfused", "cfg4_bucketed"):
            torch = self.torch
            nr = N_RAYS_PER_GPU                                    # weak: every rank traces its own 32 Mi rays
            res = int(round(nr ** 0.5)); res -= res % 2
            while nr % res:
                res -= 1
            lin_x = ekc.Float32.linspace(-1.2, 1.2, res); lin_y = ekc.Float32.linspace(-1.2, 1.2, nr // res)
            grid = ekc.meshgrid(lin_x, lin_y)                      # sphere.cpp:130-131
            g = torch.Generator(device=self.dev); g.manual_seed(1234 + self.rank)
            perm_t = torch.randperm(nr, device=self.dev, generator=g).to(torch.int32)   # shard-local permutation
            perm = ekc.UInt32.map(perm_t.data_ptr(), nr)
            mask = (synth.hash_u32(self.rank * nr, nr, 5) & ekc.UInt32(3)) != ekc.UInt32(0)     # 75 % active
            grid_xy = torch.stack((grid.x.torch(), grid.y.torch()), dim=1).contiguous() \
                if workload == "cfg4_packed" else None              # {x, y} records, built once with the scene
            out["keep"] = (perm_t, perm, mask, grid, grid_xy)
            packer = None
            F, V3 = ekc.Float32, ekc.Vector3f
            light = V3(F(-1.0), F(-1.0), F(2.0))

            import ctypes
            fused_lib = ctypes.CDLL(os.path.join(ROOT, "examples", "libsphere_fused.so"))
            hits_c = ctypes.c_uint64()

            def step_fused():
                # the same program as ONE kernel: enoki::vectorize() over the reference's templated kernels
                # (examples/sphere_fused.cpp); 18 B per ray instead of ~318
                P = ctypes.c_void_p
                if workload == "cfg4_bucketed":
                    # per PIXEL instead of per ray: rays grouped by pixel bucket once, grid / image slices streamed in order;
                    # `image = full(-1)` happens in the same pass (every pixel is written: vectorize_through_fill)
                    image = F.empty(nr)
                    rc = fused_lib.sphere_through_fill_device(P(grid.x.data_ptr()), P(grid.y.data_ptr()), P(perm.data_ptr()),
                                                              P(mask.data_ptr()), ctypes.c_size_t(nr), ctypes.c_float(-1.0),
                                                              P(image.data_ptr()), ctypes.byref(hits_c))
                elif grid_xy is not None:
                    image = F.full(-1.0, nr)
                    rc = fused_lib.sphere_fused_packed_device(P(grid_xy.data_ptr()), P(perm.data_ptr()), P(mask.data_ptr()),
                                                              ctypes.c_size_t(nr), P(image.data_ptr()), ctypes.byref(hits_c))
                else:
                    image = F.full(-1.0, nr)
                    rc = fused_lib.sphere_fused_device(P(grid.x.data_ptr()), P(grid.y.data_ptr()), P(perm.data_ptr()), P(mask.data_ptr()),
                                                       ctypes.c_size_t(nr), P(image.data_ptr()), ctypes.byref(hits_c))
                assert rc == 0
                total = self.sh.exchange.add(int(hits_c.value), "sum", torch.int64)      # hit count: ONE int64 all-reduce
                self.sh.flush()
                out["y"] = ekc.Float32(float(total.item()))
                out["image"] = image

            def step_unfused():
                pp = ekc.gather(grid, perm, mask)
                o = V3(pp.x, pp.y, F(-1.0)); d = V3(F(0.0), F(0.0), F(1.0))
                a = ekc.dot(d, d)
                b = F(2.0) * ekc.dot(o, d)
                c = ekc.dot(o, o) - F(1.0)
                discrim = b * b - F(4.0) * a * c
                t = (-b + ekc.sqrt(discrim)) / (F(2.0) * a)
                hit = discrim >= F(0.0)
                pos = ekc.select(hit, o + d * t, V3(F(0.0), F(0.0), F(0.0)))
                shade = F(0.2) + ekc.max(ekc.dot(pos, light), F(0.0)) * F(90.0)
                hit = hit & mask
                image = F.full(-1.0, nr)
                ekc.scatter(image, shade, perm, hit)
                total = self.sh.exchange.add(int(ekc.count(hit)), "sum", torch.int64)
                self.sh.flush()
                out["y"] = ekc.Float32(float(total.item()))
                out["image"] = image
            step = step_unfused if workload == "cfg4_unfused" else step_fused
        else:
            a0 = synth.uniform_pm1(begin, n, 1); b0 = synth.uniform_pm1(begin, n, 3)
            packer = self.sh if ekd.active() else None

            def compute():
                out["y"] = ekc.hsum(ekc.sin(ekc.exp(ekc.fmadd(a0, x, b0))))

            def exchange(reuse=False):
                if packer:
                    if reuse and out.get("plan") is not None:
                        out["plan"].run()
                    else:
                        out["reduced"] = [self.sh.reduce(out["y"])]
                        out["plan"] = self.sh.flush()
        ek.hip_sync()
        if workload in ("cfg3b", "cfg3a", "cfg2") or workload in CFG3B_VARIANTS:
            def step():
                compute()
                exchange()
            return step, packer, out, compute, exchange
        return step, packer, out, None, None

    def stream_ceiling(self):
        """What a plain streaming kernel of this library reaches on this GPU right now ('also report against a
        measured copy-kernel ceiling'): floor() over 64 Mi floats, 4 B read + 4 B written per element, HIP events.  (Not abs():
        the library leaves abs / neg / sin ... unevaluated until they are consumed -- there would be nothing to time.)"""
        from enoki_amd import hiprt
        n = 1 << 26
        x = self.synth.uniform_pm1(0, n, 9)
        ms = min(hiprt.time_region(self.ekc.hip_stream(), lambda: self.ekc.floor(x), iters=10, warmup=2) for _ in range(3))
        gbs = 8.0 * n / ms / 1e6
        return {"kernel": "floor, 64 Mi f32 (8 B/elt)", "GB/s": round(gbs, 1), "frac_of_peak": round(gbs / (HBM_PEAK_TBS * 1000), 4)}

    def run(self, workload, steps, warmup, profile_steps, max_seconds=None, pre_warm_s=0.0):
        torch, ek, ekd = self.torch, self.ek, self.ekd
        step, packer, out, compute, exchange = self.make_step(workload)
        self.pre_warm_s = 0.0
        if pre_warm_s > 0:
            # The driver times 20 steps (7 ms) of a process that has just started: memory clocks, the allocator's pools and
            # the instruction caches are those of an idle GPU, and the same kernels ran 12 % faster one workload later in the
            # same process (round 4).  The headline step runs for `pre_warm_s` seconds first -- outside the timed region,
            # reported -- so that the timed steps are those of a long run; --warmup / --steps stay exactly as given.
            # (the step contains the exchange: every rank must run the SAME number of steps -- the count is derived from a timed
            # batch whose duration all ranks agree on, never from a rank's own clock inside the loop)
            def batch(k):
                ekd.barrier(); torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(k):
                    step()
                if packer:
                    packer.wait_all()
                torch.cuda.synchronize()
                return ekd.max_over_ranks(time.perf_counter() - t)
            t0 = time.perf_counter()
            done = batch(20)                         # (the first steps of a process are slow: kernels load, pools fill)
            per_step, rounds = done / 20, 0
            while done < pre_warm_s and rounds < 8:
                k = int(min(max(1.0, (pre_warm_s - done) / max(per_step, 1e-6)), 100000))
                t = batch(k)
                done, per_step, rounds = done + t, t / k, rounds + 1
            self.pre_warm_s = round(time.perf_counter() - t0, 3)
        for _ in range(warmup):
            step()
        if packer:
            packer.wait_all()
        if max_seconds is not None:
            # secondary workloads: as many steps as fit the time budget (at least 5), from the duration of two more steps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(); step()
            if packer:
                packer.wait_all()
            torch.cuda.synchronize()
            steps = max(5, min(steps, int(max_seconds / max((time.perf_counter() - t0) / 2, 1e-6))))
        # (--graph) The timed steps replay a step graph: the launches of ONE forward + backward() captured on the library stream
        # (ek_hip_graph_*), so a step costs no host work (tape walk, allocator, ~12 launch calls).  The collective stays
        # outside the graph.  Workloads that read back to the host inside the step (cfg4: count) run eagerly.
        graph, replay = None, "eager (python-driven steps: tape walk, allocator, one launch call per kernel)"
        if compute is not None and self.args.graph:
            try:
                ek.hip_sync()
                ek.hip_graph_begin()
                try:
                    compute()
                finally:
                    graph = ek.hip_graph_end()
                replay = f"hipGraph ({ek.hip_graph_launch_count(graph)} kernel launches per replay)"
                out["plan"] = None                    # the exchange plan is rebuilt on the captured buffers, then reused
                timed_step = lambda: (ek.hip_graph_launch(graph), exchange(True))
                timed_step()                          # first replay outside the timed region (graph upload)
            except Exception as e:                    # never let the replay machinery break the measurement
                print(f"[bench] step graph unavailable ({type(e).__name__}: {e}); timing eager steps", file=sys.stderr)
                if graph is not None:                 # a partial capture: give its pool back
                    try:
                        ek.hip_graph_destroy(graph)
                    except Exception:
                        pass
                graph, replay, timed_step = None, "eager (the step cannot be captured: " + str(e).split(":")[0][:60] + ")", step
        else:
            timed_step = step
        if packer:
            packer.wait_all()
        ekd.barrier(); torch.cuda.synchronize()
        coll0 = self.sh.exchange.collectives
        t0 = time.perf_counter()
        for _ in range(steps):
            timed_step()
        if packer:
            packer.wait_all()             # every step's all-reduce completes inside the timed region
        coll_per_step = (self.sh.exchange.collectives - coll0) / max(steps, 1)
        torch.cuda.synchronize(); ekd.barrier()
        elapsed = ekd.max_over_ranks(time.perf_counter() - t0)
        ms_per_step = elapsed / steps * 1e3
        units = N_RAYS_PER_GPU * self.world if workload.startswith("cfg4") else N_PATHS_PER_GPU * self.world if workload.startswith("cfg5") else self.N
        graph_ms = ms_per_step if graph is not None else None

        eager_ms = ms_per_step
        if graph is not None:
            ek.hip_graph_destroy(graph)
            # the same K steps driven from python (tape walk, allocator, one launch call per kernel), same timing protocol
            step()
            if packer:
                packer.wait_all()
            ekd.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            if packer:
                packer.wait_all()
            torch.cuda.synchronize(); ekd.barrier()
            eager_ms = ekd.max_over_ranks(time.perf_counter() - t0) / steps * 1e3
            # `value` is ALWAYS the python-driven protocol (what a caller of the library gets); the replay time is reported next to it
            ms_per_step = eager_ms
            replay = "eager (python-driven steps: tape walk, allocator, one launch call per kernel); graph_ms_per_step is the hipGraph replay of the same step"
        gelem_s = units / (ms_per_step * 1e-3) / 1e9
        # per-kernel timing of the same step (run eagerly): one HIP event per launch on the library stream
        torch.cuda.synchronize()
        t_prof = time.perf_counter()
        ek.hip_profile_begin()
        for _ in range(profile_steps):
            step()
        prof = json.loads(ek.hip_profile_end())
        if packer:
            packer.wait_all()
        torch.cuda.synchronize()
        profiled_ms_per_step = (time.perf_counter() - t_prof) / max(profile_steps, 1) * 1e3     # (the steps the event deltas below decompose)
        kernels = []
        for k in prof:
            if k["launches"] == 0 or k["elements"] // k["launches"] < 1024:
                continue        # scalar bookkeeping launches (size-1 arrays) are not bandwidth kernels
            avg_ms = k["total_ms"] / k["launches"]
            bpl = k["bytes"] / k["launches"]
            kernels.append({"kernel": k["kernel"], "launches_per_step": k["launches"] / profile_steps,
                            "avg_ms": round(avg_ms, 5), "bytes_per_launch": int(bpl),
                            "tb_s": round(bpl / avg_ms / 1e9, 4) if avg_ms > 0 else None,
                            "share_ms_per_step": round(k["total_ms"] / profile_steps, 5)})
        kernels.sort(key=lambda k: -k["share_ms_per_step"])
        total_bytes_step = sum(k["bytes"] for k in prof) / profile_steps
        roofline = None
        if kernels:
            dom = kernels[0]
            achieved = dom["bytes_per_launch"] / dom["avg_ms"] / 1e6      # GB/s
            whole = total_bytes_step / (ms_per_step * 1e-3) / 1e9
            traffic, traffic_source = pmc_traffic(dom["kernel"]) if self.n == (1 << 26) else (None, "PMC summaries are taken at 64 Mi elements per GPU")
            rp_us, rp_src = rocprof_avg_us(dom["kernel"]) if self.n == (1 << 26) else (None, "kernel-trace summaries are taken at 64 Mi elements per GPU")
            frac_live = achieved / (HBM_PEAK_TBS * 1000)
            frac_rocprof = dom["bytes_per_launch"] / (rp_us * 1e-6) / 1e9 / (HBM_PEAK_TBS * 1000) if rp_us else None
            # frac: the PROFILE-derived figure -- the algorithmic bytes of a launch over the kernel's average duration in the committed
            # rocprofv3 kernel trace of THIS tree (no gaps; only when the trace's kernels_sha16 stamp equals this tree's) -- and the live
            # one otherwise; frac_live: from the HIP-event deltas of this run (a delta includes the gap to the previous launch)
            frac = frac_rocprof if frac_rocprof else frac_live
            if frac_rocprof:
                achieved = dom["bytes_per_launch"] / (rp_us * 1e-6) / 1e9
            roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_TBS * 1000, "unit": "GB/s", "frac": round(frac, 4),
                        "frac_source": "rocprofv3 kernel trace of this tree (" + KSTATS_FILE + ")" if frac_rocprof else "live HIP-event deltas (no stamped trace of this tree)",
                        "frac_live": round(frac_live, 4), "frac_rocprof": round(frac_rocprof, 4) if frac_rocprof else None,
                        "rocprof_avg_us": rp_us, "rocprof_source": rp_src,
                        "traffic": traffic, "traffic_source": traffic_source,
                        "whole_step": {"algorithmic_bytes": int(total_bytes_step),
                                       "bytes_per_elt": round(total_bytes_step / max(N_RAYS_PER_GPU if workload.startswith("cfg4") else N_PATHS_PER_GPU if workload.startswith("cfg5") else self.n, 1), 2),
                                       "achieved_GBs": round(whole, 1), "frac": round(whole / (HBM_PEAK_TBS * 1000), 4)},
                        "kernels": kernels}
        if roofline and workload == self.args.workload:
            roofline["measured_stream_ceiling"] = self.stream_ceiling()
        y_val = float(out["y"].numpy()[0])
        if packer and out.get("reduced"):
            y_val = float(out["reduced"][0].item())
        if self.args.dump_gradients and workload == "cfg3b" and workload == self.args.workload and not getattr(self, "side_record", False):
            os.makedirs(self.args.dump_gradients, exist_ok=True)
            if packer and out.get("reduced") and len(out["reduced"]) == 3:
                hA, hB = out["reduced"][1], out["reduced"][2]
                owned = hA.owned if hA.scattered else (0, K_TABLE)
                np.savez(os.path.join(self.args.dump_gradients, f"grad_rank{self.rank}.npz"), gA=hA.tensor().cpu().numpy(),
                         gB=hB.tensor().cpu().numpy(), begin=owned[0], end=owned[1])
            else:
                np.savez(os.path.join(self.args.dump_gradients, f"grad_rank{self.rank}.npz"), gA=out["gA"].numpy(), gB=out["gB"].numpy(),
                         begin=0, end=K_TABLE)
        outputs = {k: out[k].numpy() for k in ("gA", "gB", "ga", "gb") if k in out} if self.world == 1 and workload == self.args.workload else {}
        outputs["y"] = y_val
        if roofline:
            # the per-kernel times and the step they are said to add up to, taken in the SAME state of the device: kernels on
            # one in-order stream do not overlap, so sum(live event deltas, which include the launch gaps) ~ the step; the
            # committed rocprofv3 trace (no gaps) must not exceed it
            live_sum = sum(k["total_ms"] for k in prof) / profile_steps
            roofline["trace_check"] = trace_check(ms_per_step, live_sum, self.n == (1 << 26) and workload == "cfg3b", profiled_ms_per_step)
        return {"value": round(gelem_s, 3), "ms_per_step": round(ms_per_step, 4), "eager_ms_per_step": round(eager_ms, 4),
                "graph_ms_per_step": round(graph_ms, 4) if graph_ms is not None else None, "result_y": y_val,
                "roofline": roofline, "collectives_per_step": coll_per_step,
                "outputs": outputs, "replay": replay}


def cpu_baseline(workload, N):
    """Time the reference's CPU path (oracle/_ref = the unmodified reference headers; falls back to the C
    restatement) on a bounded sample: the SAME workload at N elements, repeated until ~10 s of CPU work."""
    import oracle_lib as ol
    from conftest import hash_u32, uniform_pm1
    try:
        chk, kind = ol.ref(), "reference"
    except Exception:
        chk, kind = ol.port(), "port"
    n = min(N, 1 << 26)
    x = uniform_pm1(n, 2)
    runs, t_total, t_best = 0, 0.0, None
    ref_out = {}
    if workload == "cfg3b":
        A, B = uniform_pm1(K_TABLE, 6), uniform_pm1(K_TABLE, 7)
        idx = (hash_u32(np.arange(n, dtype=np.uint64), 4) % np.uint32(K_TABLE)).astype(np.uint32)
        def fn():
            ref_out["y"], ref_out["gA"], ref_out["gB"], t = chk.cfg3b(A, B, x, idx)
            return t
    elif workload == "cfg3a":
        a, b = uniform_pm1(n, 1), uniform_pm1(n, 3)

        def fn():
            ref_out["y"], ref_out["ga"], ref_out["gb"], t = chk.cfg3a(a, x, b)
            return t
    elif workload in ("cfg4", "cfg4_packed", "cfg4_unfused", "cfg4_bucketed"):
        import ctypes
        n = 1 << 22                                   # bounded sample: 4 Mi rays of the same program
        res = 2048
        lin = np.linspace(-1.2, 1.2, res, dtype=np.float32)
        gx, gy = np.tile(lin, res), np.repeat(lin, res)
        perm = np.random.default_rng(0).permutation(n).astype(np.uint32)
        mask = ((hash_u32(np.arange(n, dtype=np.uint64), 5) & np.uint32(3)) != 0).astype(np.uint8)
        img = np.empty(n, np.float32); hc = ctypes.c_uint64()
        cfg4 = chk.lib.ref_cfg4 if kind == "reference" else chk.lib.orc_cfg4
        ptr = lambda v: v.ctypes.data_as(ctypes.c_void_p)

        def fn():
            img.fill(-1.0)
            t0 = time.perf_counter()
            cfg4(ptr(gx), ptr(gy), ptr(perm), ptr(mask), ctypes.c_size_t(n), ptr(img), ctypes.byref(hc))
            return time.perf_counter() - t0
    else:
        a, b = uniform_pm1(n, 1), uniform_pm1(n, 3)

        def fn():
            ref_out["y"], t = chk.cfg2(a, x, b)
            return t
    while runs < 2 or (t_total < 10.0 and runs < 8):
        t = fn()
        runs += 1; t_total += t
        t_best = t if t_best is None else min(t_best, t)
    result = {"value": round(n / t_best / 1e9, 5), "unit": "Gelem/s", "cores": 1, "kind": kind,
              "sample": f"{workload} at n={n} elements, best of {runs} runs ({t_total:.1f} s of CPU work), "
                        f"timed region = forward + backward() inside the checker, single thread",
              "nproc": os.cpu_count()}
    if workload in ("cfg3b", "cfg3a", "cfg2"):
        result["all_cores"] = cpu_all_cores(workload, n, kind)
    return result, (ref_out if n == N else None)


def hsum_depth(n):
    """longest chain of fp additions behind one output of the library's hsum: 1024 workgroups x 256 lanes x 4 accumulators
    in stage 1 (csrc/reduce.hip), each summing ceil(n / 2^20) entries in sequence, then 2 + 6 + 2 tree levels, then stage 2
    over the 1024 partials (4 in sequence + 2 + 6 + 2)"""
    return -(-n // (1 << 20)) + 10 + 14


def check_parity(workload, N, gpu, ref, kind):
    """The timed step's outputs against the CPU checker run on the SAME inputs at the SAME size, and against a float64
    evaluation (numpy).  Classes of SURVEY 8c: gradients of cfg3a are vertical ops -> bit-exact (A); hsum and fp
    scatter_add are order dependent (D): |gpu - f64| <= 2^-24 * (depth * sum|terms| + 4 * #terms) with the depth of OUR
    summation order; the second part is the rounding of the f32 terms themselves."""
    from conftest import hash_u32, uniform_pm1
    eps = 2.0 ** -24
    x = uniform_pm1(N, 2).astype(np.float64)
    rep = {"checked_against": f"oracle ({kind}) and float64 numpy, same inputs, n = {N}"}
    if workload == "cfg3b":
        idx = (hash_u32(np.arange(N, dtype=np.uint64), 4) % np.uint32(K_TABLE)).astype(np.int64)
        u = uniform_pm1(K_TABLE, 6).astype(np.float64)[idx] * x + uniform_pm1(K_TABLE, 7).astype(np.float64)[idx]
    else:
        u = uniform_pm1(N, 1).astype(np.float64) * x + uniform_pm1(N, 3).astype(np.float64)
    if workload == "cfg2":
        u = np.exp(u)
    s = np.sin(u)
    y64, sum_abs = float(s.sum()), float(np.abs(s).sum())
    # the f32 terms are within 4 * eps ABSOLUTE of sin(u) (32 * eps behind exp): see tests/conftest.py cfg3b_truth
    y_bound = eps * (hsum_depth(N) * sum_abs + (32 if workload == "cfg2" else 4) * N)
    from conftest import stat_sum_bound
    y_stat = stat_sum_bound(s, hsum_depth(N)) * (8.0 if workload == "cfg2" else 1.0)      # 5 sigma of independent roundings
    rep["y"] = {"gpu": gpu["y"], "oracle": float(ref["y"]), "float64": y64, "abs_err_gpu": abs(gpu["y"] - y64),
                "abs_err_oracle": abs(float(ref["y"]) - y64), "bound": y_bound, "bound_5_sigma": y_stat}
    ok = abs(gpu["y"] - y64) <= y_bound and abs(gpu["y"] - y64) <= y_stat
    if workload == "cfg3a":
        for g in ("ga", "gb"):
            same = bool(np.array_equal(gpu[g].view(np.uint32), ref[g].view(np.uint32)))
            rep[g] = {"bit_exact": same}
            ok = ok and same
    elif workload == "cfg3b":
        c = np.cos(u)
        cnt = np.bincount(idx, minlength=K_TABLE)
        for g, terms in (("gA", c * x), ("gB", c)):
            g64 = np.bincount(idx, weights=terms, minlength=K_TABLE)
            bound = eps * (cnt * np.bincount(idx, weights=np.abs(terms), minlength=K_TABLE) + 4 * cnt)
            err_gpu, err_ref = np.abs(gpu[g] - g64), np.abs(ref[g] - g64)
            rep[g] = {"max_abs_err_gpu": float(err_gpu.max()), "max_abs_err_oracle": float(err_ref.max()),
                      "max_err_over_bound": float((err_gpu / np.maximum(bound, 1e-30)).max()),
                      "max_abs_diff_gpu_oracle": float(np.abs(gpu[g] - ref[g]).max())}
            ok = ok and bool(np.all(err_gpu <= bound))
    rep["parity_checked"] = bool(ok)
    return rep


def _cpu_shard_worker(job):
    """one index-range shard of the workload on one host core (spawned process: no torch, no HIP), pinned to that core;
    inputs are generated (= touched) before the first run, one untimed warm-up run faults the checker's own buffers in,
    then the MEDIAN of three timed runs counts"""
    workload, kind, begin, count, core = job
    try:
        os.sched_setaffinity(0, {core})
    except Exception:
        pass
    import oracle_lib as ol
    from conftest import hash_u32, uniform_pm1
    chk = ol.ref() if kind == "reference" else ol.port()
    i = np.arange(begin, begin + count, dtype=np.uint64)

    def u(seed):
        h = hash_u32(i, seed)
        return ((h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) * np.float32(2.0) + np.float32(-1.0)).astype(np.float32)
    x = u(2)
    if workload == "cfg3b":
        A, B = uniform_pm1(K_TABLE, 6), uniform_pm1(K_TABLE, 7)
        idx = (hash_u32(i, 4) % np.uint32(K_TABLE)).astype(np.uint32)
        run = lambda: chk.cfg3b(A, B, x, idx)[-1]
    elif workload == "cfg3a":
        a, b = u(1), u(3)
        run = lambda: chk.cfg3a(a, x, b)[-1]
    else:
        a, b = u(1), u(3)
        run = lambda: chk.cfg2(a, x, b)[-1]
    run()
    return sorted(run() for _ in range(3))[1]


def cpu_all_cores(workload, n, kind):
    """OUR index-range split of the reference's single-threaded CPU path over the host cores (Enoki itself has no
    threading): P pinned processes, each runs the checker on its own n/P shard (warm-up + median of 3); rate = n / slowest
    shard.  Clearly not a number of the reference."""
    import concurrent.futures
    import multiprocessing
    try:
        cores = sorted(os.sched_getaffinity(0))
    except Exception:
        cores = list(range(os.cpu_count() or 2))
    procs = max(1, min(64, len(cores) // 2))
    cores = cores[::max(1, len(cores) // procs)][:procs]       # spread over the sockets / SMT pairs the affinity mask offers
    per = n // procs
    jobs = [(workload, kind, r * per, per, cores[r]) for r in range(procs)]
    try:
        with concurrent.futures.ProcessPoolExecutor(procs, mp_context=multiprocessing.get_context("spawn")) as ex:
            times = list(ex.map(_cpu_shard_worker, jobs, timeout=240))
        return {"value": round(per * procs / max(times) / 1e9, 4), "unit": "Gelem/s", "cores": procs,
                "note": "our index-range split of the reference's CPU path over pinned host cores (warm-up, median of 3 per shard, "
                        "slowest shard counts); not a reference number"}
    except Exception as e:          # never let the side measurement break the bench line
        return {"value": None, "cores": procs, "note": f"all-cores measurement failed: {type(e).__name__}: {e}"}


def relaunch_as_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command line (torch.distributed.run, one rank per
    GPU, rendezvous on 127.0.0.1) and hand their exit code back.  Fewer than N visible GPUs: the ranks share them and the
    collectives go through gloo (RCCL refuses two ranks on one device) -- the line then says so (`config.backend`)."""
    import socket
    import subprocess
    import torch
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    if torch.cuda.device_count() < args.gpus:
        env.setdefault("ENOKI_DIST_BACKEND", "gloo")
        print(f"[bench] --gpus {args.gpus} on {torch.cuda.device_count()} visible GPU(s): ranks share devices, collectives through gloo",
              file=sys.stderr)
    env["ENOKI_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus)]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    launched = "RANK" in os.environ or int(os.environ.get("WORLD_SIZE", "1")) > 1
    if args.gpus > 1 and not launched:
        sys.exit(relaunch_as_ranks(args))
    if launched:
        # A multi-process run must never sit in a collective forever (a rank that died, a rendezvous that never
        # completes): after 15 minutes every rank dumps its python stack and exits non-zero instead of hanging.
        import faulthandler
        faulthandler.dump_traceback_later(900, exit=True)
    b = Bench(args)
    main_res = b.run(args.workload, args.steps, args.warmup, args.profile_steps, pre_warm_s=args.pre_warm_s)
    pre_warm_s = b.pre_warm_s
    # N > 1 under the contract's (strong) mode: the weak measurement of the SAME run as a labelled sub-record -- every rank takes
    # part (collectives), rank 0 reports
    weak = None
    if b.world > 1 and not b.weak and not args.no_weak:
        bw = Bench(args, scaling="weak")
        bw.side_record = True               # (no gradient dump: --dump-gradients is about the record's problem)
        rw = bw.run(args.workload, args.steps, args.warmup, 1, pre_warm_s=min(args.pre_warm_s, 0.3))
        weak = {"scaling": "weak", "value": rw["value"], "unit": "Gelem/s", "ms_per_step": rw["ms_per_step"],
                "elements_per_gpu": bw.n, "elements_total": bw.N,
                "note": "every GPU owns --n elements of an array of N x --n; all elements of all ranks / max-over-ranks time; same kernels, same exchange"}
        del bw
    also = {}
    if b.world == 1 and not args.no_also:
        for w in ("cfg3a", "cfg2", "cfg3b") + tuple(CFG3B_VARIANTS) + ("cfg4_bucketed", "cfg4", "cfg4_packed", "cfg4_unfused", "cfg5", "cfg5_unfused"):
            if w != args.workload:
                r = b.run(w, args.steps // 2, 2, 3, max_seconds=1.0)
                also[w] = {"value": r["value"], "unit": "Gelem/s", "ms_per_step": r["ms_per_step"],
                           "whole_step_frac_of_hbm_peak": r["roofline"]["whole_step"]["frac"] if r["roofline"] else None,
                           "bytes_per_elt": r["roofline"]["whole_step"]["bytes_per_elt"] if r["roofline"] else None,
                           "dominant_kernel": r["roofline"]["kernel"] if r["roofline"] else None,
                           "dominant_kernel_frac": r["roofline"]["frac"] if r["roofline"] else None,
                           "workload": DESCRIPTION[w]}
                if w in CFG3B_VARIANTS:
                    also[w]["parity"] = "against the reference build and float64 at 64 Mi elements: tests/test_headline_parity_gpu.py::test_cfg3b_neighbours_at_the_headline_size"
                if w.startswith("cfg5"):
                    also[w]["unit"] = "G paths/s"
                    also[w]["parity"] = ("against examples/path_trace.h instantiated on the reference's arrays (oracle/_ref: ref_cfg5), 1 Mi paths, "
                                         "loss and texture gradient inside the class-D bounds: tests/test_cfg5_gpu.py")
    cpu, parity = None, None
    if b.rank == 0 and b.world == 1 and not args.no_cpu_baseline:
        cpu, ref_out = cpu_baseline(args.workload, b.N)
        if ref_out and args.workload in ("cfg3b", "cfg3a", "cfg2"):
            parity = check_parity(args.workload, b.N, main_res["outputs"], ref_out, cpu["kind"])
            if not parity["parity_checked"]:
                print(f"[bench] PARITY FAILURE: {json.dumps(parity)}", file=sys.stderr)
    if b.rank == 0:
        line = {
            "metric": "Gelem/s + %HBM-roofline, 64M-elt DiffArray backward(), 1/2/4/8 MI355X",
            "value": main_res["value"], "unit": "Gelem/s", "n_gpus": b.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "pre_warm_s": pre_warm_s, "value_source": "eager", "eager_ms_per_step": main_res["eager_ms_per_step"],
            "graph_ms_per_step": main_res["graph_ms_per_step"], "higher_is_better": True,
            "scaling": "weak" if b.weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {DESCRIPTION[args.workload]}",
                       "elements_total": (N_RAYS_PER_GPU if args.workload.startswith("cfg4") else N_PATHS_PER_GPU) * b.world
                       if args.workload in ("cfg4", "cfg4_packed", "cfg4_unfused", "cfg4_bucketed", "cfg5") else b.N,
                       "elements_per_gpu": N_RAYS_PER_GPU if args.workload.startswith("cfg4") else N_PATHS_PER_GPU if args.workload == "cfg5" else b.n, "table_size": K_TABLE if args.workload == "cfg3b" else None,
                       "sharding": f"index-range x{b.world}", "world_size": b.group_world(), "backend": b.backend(),
                       "collectives_per_step": main_res["collectives_per_step"],
                       "gradient_exchange": (None if not b.ekd.active() or args.workload != "cfg3b" else
                                             "ONE all-reduce per step: the loss and both gradient tables, every rank holds all K bins" if not args.reduce_scatter_grads else
                                             "ONE reduce-scatter per step: rank r receives bins [r K / P, (r + 1) K / P) of both tables, the loss rides in an extra column"),
                       "step_replay": main_res["replay"]},
            "result_y": main_res["result_y"], "parity_checked": bool(parity and parity["parity_checked"]), "parity": parity,
            "roofline": main_res["roofline"], "cpu_baseline": cpu, "also": also or None, "weak": weak,
        }
        print(json.dumps(line), flush=True)
    if b.ekd.active():
        # the measurement is complete and printed: tear the group down, but do not let a stuck teardown hold the job
        import faulthandler
        import threading
        import torch.distributed as dist
        faulthandler.cancel_dump_traceback_later()
        killer = threading.Timer(120.0, lambda: os._exit(0))
        killer.daemon = True
        killer.start()
        b.ekd.barrier()
        dist.destroy_process_group()
        killer.cancel()


if __name__ == "__main__":
    main()
