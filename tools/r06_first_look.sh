#!/bin/bash
# round 6, first look: the step at the reference's own latent width (default.ini / kelsey_iterable.ini: L = 256) and at
# large batches, with the kernels as round 5 left them; rocprofv3 kernel stats of the L = 256 step.
set -e
export TMPDIR=/tmp
O=gpurun_out/r06_first
mkdir -p $O
for shape in "1024 2048 64 4096" "1024 2048 256 4096" "1024 2048 64 32768" "1024 2048 256 32768" "1024 2048 64 131072" "1024 2048 256 131072"; do
  python tools/step_time.py --shape $shape --steps 200 --reps 3 --graph >> $O/steps.txt 2>&1
done
cat $O/steps.txt
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof256 -o l256 --output-format csv -- python3 $R/tools/step_time.py --shape 1024 2048 256 4096 --steps 100 --reps 1 > $R/$O/prof256.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof256big -o l256big --output-format csv -- python3 $R/tools/step_time.py --shape 1024 2048 256 131072 --steps 100 --reps 1 > $R/$O/prof256big.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O/prof64big -o l64big --output-format csv -- python3 $R/tools/step_time.py --shape 1024 2048 64 131072 --steps 100 --reps 1 > $R/$O/prof64big.log 2>&1
cd $R
for d in prof256 prof256big prof64big; do echo "== $d"; f=$(find $O/$d -name '*kernel_stats.csv' | head -1); head -16 $f | cut -c1-200; done
