#!/bin/bash
# kernel resource usage of one csrc/*.hip: name, VGPRs, AGPRs, spills, scratch, LDS, code bytes.  usage: scripts/kres.sh ktx_linear [filter]
f=$1; pat=${2:-.}
mkdir -p /tmp/kres && cd /tmp/kres
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value --cuda-device-only -c /root/repo/ktransformers_amd/csrc/$f.hip -o $f.co -Rpass-analysis=kernel-resource-usage 2> $f.rem
python3 - "$f.rem" "$pat" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
cur = None; rows = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-z/A-Z]+\])?: (\S+) \[-Rpass", line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, nm in zip(rows, names):
    if not pat.search(nm): continue
    r = rows[mangled]
    print(f"{nm[:90]:90s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} spill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize','?'):>4s} occ {r.get('Occupancy','?'):>2s} lds {r.get('LDS Size','?')}")
PY
