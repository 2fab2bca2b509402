cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/sk -o t -- python /root/repo/tools/diag/step_kernels.py 12 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/root/repo/gpurun_out/sk/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","") for r in rows]
# markers: fill kernels with grid matching 7777 -> find 'vectorized_elementwise_kernel' FillFunctor
marks=[i for i,n in enumerate(names) if "FillFunctor" in n]
print(len(names), len(marks))
a,b=marks[-3],marks[-2]
for n,r in zip(names[a+1:b], rows[a+1:b]):
    print(n[:70], round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,1))
PY
