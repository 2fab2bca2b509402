python -m nextbestpath_amd.build > /dev/null 2>&1
for t in 6; do echo "=== tile $t"; python tools/bench_conv.py --tile $t 2>&1 | grep -v amdgpu | tail -16; done
echo "=== tile 6 batch 8"; python tools/bench_conv.py --tile 6 --batch 8 --reps 10 2>&1 | grep -v amdgpu | tail -16
echo "=== tile 6 split sweep Up5"; for sk in 4 8 16; do python tools/bench_conv.py --tile 6 --split $sk --only Up5 2>&1 | grep Up5; done
echo "=== tile 1 split sweep Up5"; for sk in 8 16 32; do python tools/bench_conv.py --tile 1 --split $sk --only Up5 2>&1 | grep Up5; done
