"""Register / scratch / LDS use of every kernel in libenoki-hip.so (hipcc -Rpass-analysis=kernel-resource-usage): anything that
spills or touches scratch memory is listed first.  python tools/kernel_resources.py [file.hip ...] > profiles/kernel_resources_rNN.txt"""
import concurrent.futures, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enoki_amd import _build as B

srcs = sys.argv[1:] or [os.path.join(B.CSRC, s) for s in B.LIB_SOURCES if s.endswith(".hip")]


def one(src):
    cmd = [B.HIPCC] + B.DEVICE + B.COMMON + B.FILE_FLAGS.get(os.path.basename(src), []) + ["--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: \s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = {"name": m.group(2), "file": os.path.basename(src)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(1).split(" ")[0] + (" Spill" if "Spill" in m.group(1) else "")] = int(m.group(2))
    return rows


with concurrent.futures.ThreadPoolExecutor(8) as ex:
    rows = [r for rs in ex.map(one, srcs) for r in rs]
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, demangle):
    r["pretty"] = re.sub(r"\(.*", "", d.replace("void ", "").replace("(anonymous namespace)::", ""))[:110]
bad = [r for r in rows if r.get("ScratchSize", 0) or r.get("VGPRs Spill", 0) or r.get("SGPRs Spill", 0)]
print(f"# {len(rows)} kernels in {len(srcs)} files; {len(bad)} use scratch memory or spill")
print(f"# {'file':20s} {'VGPR':>4s} {'AGPR':>4s} {'occ':>3s} {'LDS':>6s} {'scratch':>7s} {'vspill':>6s} {'sspill':>6s}  kernel")
for r in sorted(rows, key=lambda r: (-(r.get("ScratchSize", 0) + r.get("VGPRs Spill", 0)), r["file"], r["pretty"])):
    if r in bad or "--all" in os.environ.get("KR_FLAGS", "--all"):
        print(f"  {r['file']:20s} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('Occupancy', 0):3d} {r.get('LDS', 0):6d} "
              f"{r.get('ScratchSize', 0):7d} {r.get('VGPRs Spill', 0):6d} {r.get('SGPRs Spill', 0):6d}  {r['pretty']}")
