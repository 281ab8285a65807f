"""Profile artefact: HBM traffic of the routed-expert decode kernels per weight format, from the PMC counters, beside the
algorithmic bytes.  Two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE; counters only + kernel trace) over
scripts/moe_formats_bench.py; per kernel NAME the mean over its dispatches (MI355X_MICROARCH.md, HBM section: KiB units,
FETCH_SIZE x2 for the wide coalesced reads of gfx950, WRITE_SIZE as reported).

    python scripts/formats_pmc.py [--formats RAWINT4,q4_k_m,IQ1_S,AMXINT4] [--out gpurun_out/formats_pmc.txt]
"""
import argparse, csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--formats", default="AMXINT4,RAWINT4,q4_k_m,IQ1_S")
ap.add_argument("--layers", type=int, default=3)
ap.add_argument("--out", default="")
args = ap.parse_args()
H, I, k = 7168, 2048, 8
BPW = {"AMXINT4": 0.5 + 4 / 7168 / 2, "RAWINT4": 0.5 + 2 / 32, "q4_k_m gate/up": 144 / 256, "q4_k_m down": 210 / 256, "IQ1_S": 50 / 256}
rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix=f"ktx_fpmc_{counter}_", dir="/tmp")
    cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, "scripts", "moe_formats_bench.py"), "--formats", args.formats, "--layers", str(args.layers), "--experts", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter file:", (r.stderr or r.stdout)[-400:]); sys.exit(1)
    for x in csv.DictReader(open(files[0])):
        if x["Counter_Name"] == counter and "moe_dec" in x["Kernel_Name"]:
            a = res.setdefault(x["Kernel_Name"], {}).setdefault(counter, [0, 0.0])
            a[0] += 1; a[1] += float(x["Counter_Value"])
    shutil.rmtree(d, ignore_errors=True)
lines = [f"routed-expert decode kernels, one token, top-{k} experts of {H} x {I} (V3 / R1 / K2 layer shape): HBM traffic from the PMC counters",
         "algorithmic MB: gate|up = 8*2*H*I*bpw, down = 8*H*I*bpw (AMXINT4 0.5 + scales, RAWINT4 0.5625, Q4_K 0.5625, Q6_K 0.8203, IQ1_S 0.1953 B per weight)",
         f"{'dispatches':>10s} {'read MB':>9s} {'write MB':>9s}  kernel"]
for name, c in sorted(res.items()):
    n = c.get("FETCH_SIZE", [0, 0])[0]
    rd = c.get("FETCH_SIZE", [1, 0.0]); wr = c.get("WRITE_SIZE", [1, 0.0])
    lines.append(f"{n:10d} {rd[1] / max(rd[0], 1) * 1024 * 2 / 1e6:9.2f} {wr[1] / max(wr[0], 1) * 1024 / 1e6:9.3f}  {name[:150]}")
for nm, b in BPW.items():
    lines.append(f"algorithmic {nm:16s}: gate|up {k * 2 * H * I * b / 1e6:7.2f} MB   down {k * H * I * b / 1e6:7.2f} MB")
txt = "\n".join(lines)
print(txt)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write(txt + "\n")
