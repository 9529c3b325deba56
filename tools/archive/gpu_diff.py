import sys, numpy as np
sys.path.insert(0,'.')
import rendering_amd as RA
from oracle import oracle as O
from rendering_amd import assets; assets.ensure()
name,w,h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
o = O.OracleScene('scenes/%s.scene'%name, w, h); g = RA.Scene('scenes/%s.scene'%name, w, h)
a = o.pass1(); b = g.render_host(ssaa=False)
d = (a.view(np.uint32)!=b.view(np.uint32)).any(-1)
print('ndiff', d.sum(), 'maxabs', np.nanmax(np.abs(a-b)), 'nan', np.isnan(b).sum())
ys,xs = np.nonzero(d)
for y,x in list(zip(ys,xs))[:12]: print(y,x,a[y,x],b[y,x], (a[y,x].view(np.uint32).astype(np.int64)-b[y,x].view(np.uint32).astype(np.int64)))
# classify by primary-hit object via oracle probe? print histogram of rows
print('rows', np.bincount(ys//32), 'cols', np.bincount(xs//32))
