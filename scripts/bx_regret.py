import glob, json, math, os, sys
sys.path.insert(0, '/root/repo')
import deepgemm_ascend_amd as dga
shapes = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/shape_*_rank_*.jsonl"):
        for line in open(f):
            r = json.loads(line)
            if r["negative"] or r["time"] <= 0: continue
            p = r["parameters"]
            shapes.setdefault((r["M"], r["N"], r["K"]), {})[(p["m1"], p["n1"], p["splitk"])] = r["time"]
rows=[]; miss=0
for (m,n,k),cs in shapes.items():
    t = dga.tiling(m,n,k,policy="bf16_exact")
    bm = 128 if t.m1>=128 else (64 if t.m1>=64 else 32); bn = 256 if t.n1>=256 else 128
    if (bm,bn) not in {(128,256),(128,128),(64,256),(64,128),(32,128)}: bn=128
    key=(bm,bn,max(1,t.splitkFactor))
    best=min(cs.values())
    if key not in cs: miss+=1; continue
    rows.append((cs[key]/best,(m,n,k),key,cs[key],best))
reg=[r[0] for r in rows]
print(len(rows),'shapes',miss,'missing: geomean %.4f mean %.4f p90 %.3f max %.3f'%(math.exp(sum(map(math.log,reg))/len(reg)),sum(reg)/len(reg),sorted(reg)[int(.9*len(reg))],max(reg)))
for r in sorted(rows,reverse=True)[:8]: print('  %.2f %s %s %.1f %.1f'%r)
