import sys
sys.path.insert(0,'tests')
from engine_util import EngineVec
for flags,name in ((0,'full'),(32,'no bg loads'),(64,'no tile loads'),(96,'no loads'),(2,'no sprites'),(4,'no bg/tiles'),(14,'nothing')):
    e=EngineVec('coinrun',65536,seed_base=1); e.reset(); e.timed(40)
    e.set_debug(flags)
    tot,ren=e.timed(64)
    print('%-14s render %.3f ms  total %.3f ms'%(name,ren/64,tot/64)); e.close()
