import re,sys
lines=open('/root/repo/gpurun_out/torch_prof.txt').read().split('\n')
n=int(sys.argv[1]) if len(sys.argv)>1 else 60
steps=4
rows=[]
for l in lines[3:]:
    f=re.split(r'\s{2,}',l.strip())
    if len(f)<10 or not f[-1].isdigit(): continue
    name=f[0]; selfcuda=f[6]; calls=f[-1]
    def ms(x):
        return float(x[:-2])*(1 if x.endswith('ms') else 1e-3) if x[-2:] in('ms','us') else 0
    rows.append((ms(selfcuda)/steps,int(calls)//steps,name[:70]))
for t,c,nm in rows[:n]:
    if nm.startswith('void') or '_kernel' in nm or nm.startswith('Memset') or nm.startswith('Memcpy'): continue
    print(f"{t:8.3f} ms {c:5d}  {nm}")
