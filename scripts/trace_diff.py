import sys
def load(p):
    rows={}; grids={}
    sec=0
    for l in open(p):
        if l.startswith('Per launch'): sec=1
        if not l.startswith('| `'): continue
        c=[x.strip() for x in l.strip().strip('|').split('|')]
        if sec==0: rows[c[0]]=(int(c[1]),float(c[2]))
        else: grids[(c[0],c[1])]=(int(c[2]),float(c[3]),float(c[4]))
    return rows,grids
a,ag=load(sys.argv[1]); b,bg=load(sys.argv[2])
skip=('pack','wscale','copyBuffer')
ta=sum(v[1] for k,v in a.items() if not any(s in k for s in skip))/13; tb=sum(v[1] for k,v in b.items() if not any(s in k for s in skip))/13
print(f"sum of kernels per forward: A {ta:.3f} ms  B {tb:.3f} ms")
for k in sorted(set(a)|set(b), key=lambda k:-(b.get(k,(0,0))[1])):
    x=a.get(k,(0,0))[1]/13; y=b.get(k,(0,0))[1]/13
    if max(x,y)>0.05 and not any(s in k for s in skip): print(f"  {k[:60]:60s} A {x:7.3f}  B {y:7.3f}  d {y-x:+.3f}")
print("per grid:")
for k in sorted(set(ag)|set(bg), key=lambda k:-abs(bg.get(k,(0,0,0))[2]-ag.get(k,(0,0,0))[2])):
    x=ag.get(k,(0,0,0)); y=bg.get(k,(0,0,0))
    if abs(y[2]-x[2])/13>0.01: print(f"  {k[0][:40]:40s} {k[1]:12s} n {y[0]:4d} A {x[1]:7.1f} us B {y[1]:7.1f} us  d/fwd {(y[2]-x[2])/13:+.3f} ms")
