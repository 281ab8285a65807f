"""Per-kernel in-graph durations AND the gap to the predecessor from a (trimmed) rocprofv3 kernel trace of graph replays.
usage: python scripts/trace_summary.py gpurun_out/<dir>/ktrace_tail.csv.gz [nsteps]
A decode step is delimited by argmax_bf16_kernel; the last `nsteps` complete steps are averaged position by position, then
grouped by kernel name."""
import csv
import gzip
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0][:64]


def load(path):
    op = gzip.open if path.endswith(".gz") else open
    rows = list(csv.DictReader(op(path, "rt")))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    return rows


def steps_of(rows):
    cuts = [i for i, r in enumerate(rows) if "argmax_bf16_kernel" in r["Kernel_Name"]]
    return [rows[a + 1:b + 1] for a, b in zip(cuts, cuts[1:])]


def main():
    rows = load(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    steps = steps_of(rows)
    L = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
    steps = [s for s in steps if len(s) == L][-n:]
    print(f"{len(steps)} steps of {L} dispatches; step span avg "
          f"{sum(s[-1]['e'] - s[0]['s'] for s in steps) / len(steps) / 1e3:.1f} us")
    agg = OrderedDict()
    for s in steps:
        for i, r in enumerate(s):
            k = short(r["Kernel_Name"]) + f" g{r['Grid_Size_X']}"
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += (r["e"] - r["s"]) / 1e3
            if i:
                a[2] += (r["s"] - s[i - 1]["e"]) / 1e3
    tot_d = tot_g = 0.0
    print(f"{'kernel':72s} {'n/step':>6s} {'dur us':>8s} {'gap us':>7s} {'tot/step':>9s}")
    for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        per = c / len(steps)
        print(f"{k:72s} {per:6.1f} {d / c:8.2f} {g / c:7.2f} {(d + g) / len(steps):9.1f}")
        tot_d += d / len(steps)
        tot_g += g / len(steps)
    print(f"sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us")
    # one MoE layer in order (last layer of the last step)
    s = steps[-1]
    idx = [i for i, r in enumerate(s) if "moe_dec_down" in r["Kernel_Name"]]
    if len(idx) >= 2:
        a, b = idx[-2] + 1, idx[-1] + 1
        print("--- last MoE layer of the last step, in launch order")
        for i in range(a, b):
            r = s[i]
            print(f"  {short(r['Kernel_Name']):60s} grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>4s} lds {r['LDS_Block_Size']:>6s} "
                  f"vgpr {r['VGPR_Count']:>3s}+{r['Accum_VGPR_Count']:>3s} dur {(r['e'] - r['s']) / 1e3:7.2f} gap {(r['s'] - s[i - 1]['e']) / 1e3:6.2f}")
        print(f"  layer span {(s[b - 1]['e'] - s[a - 1]['e']) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
