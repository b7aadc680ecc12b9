import sys
sys.path.insert(0,'tests')
from engine_util import EngineVec
for flags,name in ((0,'full'),(512,'rows never blend'),(128,'no row loop'),(128+2,'no rows/sprites'),(128+2+8,'no rows/spr/store'),(4+2+8,'nothing')):
    e=EngineVec('coinrun',65536,seed_base=1); e.reset(); e.timed(40)
    e.set_debug(flags)
    tot,ren=e.timed(64)
    print('%-18s render %.3f ms  total %.3f ms'%(name,ren/64,tot/64)); e.close()
