#!/bin/bash
# First thing in every measurement call of round 4: which box is this (id, driver, class by the short decode bench) and what do
# the translation / first-touch probe and the clocks under load say on it.  ~45 s.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-prelude}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
$R/scripts/tlb_probe 48 > $O/tlb_probe.txt 2>&1
bash $R/scripts/box_load_state.sh $(basename $O) > $O/load.txt 2>&1
head -1 $O/load.txt; sed -n 2p $O/load.txt | cut -c1-400
cat $O/tlb_probe.txt
