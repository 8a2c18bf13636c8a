#!/bin/bash
# Every probe behind profiles/r04_notes.md in one pass on one box (logs -> gpurun_out/r04_probes/, copied to
# profiles/r04_raw/ by hand).  Needs the tools library: hipcc ... -DAEW_FN_ABLATE=1 -o ae-wavenet_amd/lib/libaewavenet_hip_abl.so
#   usage (on the GPU box): tools/round4_probes.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r04_probes; mkdir -p $O; cd $R
ABL=ae-wavenet_amd/lib/libaewavenet_hip_abl.so      # tools build: not shipped with the snapshot (.gpurunignore), built here
[ -f $ABL ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -ffp-contract=off -Wno-unused-result -Wno-inline-asm \
    -DAEW_FN_ABLATE=1 ae-wavenet_amd/csrc/aewavenet.hip -o $ABL
python tools/membound_probe.py > $O/membound_probe.log 2>&1
AEW_LIB_PATH=$ABL python tools/phase_clock.py > $O/phase_clock.log 2>&1
AEW_LIB_PATH=$ABL python tools/overlap_probe.py > $O/overlap_plain_kernel.log 2>&1
WINDOW=64 DIL=16 ROWS192=1 python tools/overlap_probe.py > $O/overlap_window_kernel_rows6900.log 2>&1
WINDOW=64 DIL=16 ROWS192=1 ROWS=6000 python tools/overlap_probe.py > $O/overlap_window_kernel_rows6000.log 2>&1
python tools/wgrad_probe.py > $O/wgrad_probe.log 2>&1
(cd tools/ubench && ./store_pattern 768 56320 512) > $O/store_pattern.log 2>&1
python tools/ab_step.py base nt_mem128=1 nt_mem128=2 --rounds 3 --steps 20 > $O/ab_mem128.log 2>&1
python tools/ab_step.py base nt_window=0 nt_window=0,nt_deep=1 nt_window=0,nt_deep=2 nt_deep=3 --rounds 2 --steps 20 > $O/ab_deep.log 2>&1
python tools/ab_step.py base E.split_chains=1 E.split_chains=1,lanes=2 --rounds 3 --steps 30 > $O/ab_split_chains_step.log 2>&1
for pl in fwd_b bwd; do python tools/ab_step.py base E.split_chains=1,lanes=2 --rounds 3 --steps 30 --plan $pl > $O/ab_split_chains_$pl.log 2>&1; done
grep -h -v amdgpu.ids $O/ab_*.log | head -40
