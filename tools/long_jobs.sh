#!/bin/bash
# End-to-end long jobs through the CLI: C3 at its full 4 096 spp, and scene.example.json at 7680x4320 with
# 2 048 spp in checkpointed chunks, stopped half way and resumed; writes wall times, Mpaths/s and the equality check.
# usage (GPU box, repo root): bash tools/long_jobs.sh <tag>
set -u
TAG=${1:-long}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
EXE=$R/pathtracer_amd/bin/hippt
t() { python3 -c "import time; print(time.time())"; }
{
echo "== C3: scenes/glass_fog.json 1920x1080 x 4096 spp, one GPU, CLI end to end (load, render, resolve, PNG)"
T0=$(t); timeout -k 10 300 $EXE --gpu -q --scene $R/scenes/glass_fog.json --width 1920 --height 1080 --samples 4096 --seed 0 --out $OUT/c3_4096.png | grep -v "^  \|^===\|^$" ; T1=$(t)
python3 -c "w=$T1-$T0; print(f'wall {w:.2f} s  ->  {1920*1080*4096/w/1e6:.0f} Mpaths/s end to end (process start to PNG on disk)')"
ls -l $OUT/c3_4096.png | awk '{print "png bytes", $5}'
echo "== C5 geometry: scenes/scene.example.json 7680x4320 x 2048 spp in chunks of 128 with a checkpoint after every chunk"
T0=$(t); timeout -k 10 600 $EXE --gpu -q --scene $R/scenes/scene.example.json --width 7680 --height 4320 --samples 2048 --seed 0 --chunk 128 --checkpoint $OUT/c5.acc --out $OUT/c5_2048_oneshot.png | grep -v "^  \|^===\|^$" | tail -4; T1=$(t)
python3 -c "w=$T1-$T0; print(f'wall {w:.2f} s  ->  {7680*4320*2048/w/1e6:.0f} Mpaths/s end to end, 16 chunks, 16 checkpoints of 398 MB')"
cp $OUT/c5.acc $OUT/c5_full.acc
echo "== the same job stopped after 1024 spp and resumed"
rm -f $OUT/c5.acc
T0=$(t); timeout -k 10 600 $EXE --gpu -q --scene $R/scenes/scene.example.json --width 7680 --height 4320 --samples 1024 --seed 0 --chunk 128 --checkpoint $OUT/c5.acc --out $OUT/c5_half.png | tail -1; T1=$(t)
timeout -k 10 600 $EXE --gpu -q --scene $R/scenes/scene.example.json --width 7680 --height 4320 --samples 2048 --seed 0 --chunk 128 --checkpoint $OUT/c5.acc --resume --out $OUT/c5_2048_resumed.png | grep -v "^  \|^===\|^$" | tail -3; T2=$(t)
python3 -c "print(f'first half {$T1-$T0:.2f} s, resumed half {$T2-$T1:.2f} s')"
cmp $OUT/c5.acc $OUT/c5_full.acc && echo "checkpoint after resume == checkpoint of the uninterrupted job (byte for byte)"
cmp $OUT/c5_2048_oneshot.png $OUT/c5_2048_resumed.png && echo "PNG after resume == PNG of the uninterrupted job (byte for byte)"
ls -l $OUT/c5_2048_oneshot.png | awk '{print "png bytes", $5}'
} > $OUT/report.txt 2>&1
rm -f $OUT/*.acc $OUT/c5_*.png   # 400 MB each: not merged back
cat $OUT/report.txt
