python -m nextbestpath_amd.build > /dev/null 2>&1
echo "== batch 2"; python tools/bench_conv.py --sweep --batch 2 2>&1 | grep -v amdgpu
echo "== batch 8"; python tools/bench_conv.py --sweep --batch 8 2>&1 | grep -v amdgpu
