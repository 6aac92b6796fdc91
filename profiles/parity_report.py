import sys, os
sys.path.insert(0, os.getcwd())
from tests import parity as P
for name, cfg, B, T in (('cfg1', P.CFG1, 64, 100), ('cfg2', P.CFG2, 18, 130), ('cfg4', P.CFG4, 18, 130)):
    r = P.check_train(cfg, B=B, T=T)
    print(name, 'max-norm worst', max(v for v in r.values() if isinstance(v, float)))
    print('  elem', {k: f'{v:.2e}' for k, v in r['elem'].items()})
    print('  noise', r['noise'])
