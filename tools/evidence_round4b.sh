# Round-4 evidence, second part (one gpurun call): the probes DESIGN.md quotes for the fp8 weight path, the API path's host
# time and the cross-stream edge of the data-parallel step.  Writes gpurun_out/r04_*; the builder copies them to profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
hipcc --offload-arch=gfx950 -O2 tools/probe_extlaunch.hip -o $O/probe_extlaunch 2>/dev/null && timeout -k 5 120 $O/probe_extlaunch > $O/r04_extlaunch.txt 2>&1; cat $O/r04_extlaunch.txt
python tools/api_breakdown.py 2>&1 | grep -v amdgpu > $O/r04_api_breakdown.txt; cat $O/r04_api_breakdown.txt
python tools/fp8_launches.py 2>&1 | grep -v amdgpu > $O/r04_fp8_launches.txt; cat $O/r04_fp8_launches.txt
for p in 0 3 6 9 12 15 18 22 0 15; do RV_WGRAD_TAIL_PCT=$p python tools/tail_sweep.py 2>&1 | grep "tail pct"; done > $O/r04_tail_sweep.txt; cat $O/r04_tail_sweep.txt
for n in cross_stream stream_value_ops store_rate ds_read_tr8; do hipcc --offload-arch=gfx950 -O2 tools/probe_$n.hip -o $O/probe_$n 2>/dev/null && { echo "== tools/probe_$n.hip"; timeout -k 5 120 $O/probe_$n; } ; done > $O/r04_probes.txt 2>&1; cat $O/r04_probes.txt
