import sys; sys.path.insert(0,'.')
import oracle, bls12_381 as m
from oracle import clib
from blst_eip2537_amd import Eip2537Executor as X, Eip2537Error
def run(name, inp):
    try: r=(0,X.pairing(inp).hex()[-2:])
    except Eip2537Error as e: r=(e.code,None)
    print(name, r, clib.call('bls12_pairing',inp)[0], X.last_timing())
pr=lambda ps: b"".join(m.encode_g1(p)+m.encode_g2(q) for p,q in ps)
run('G1,G2', pr([(m.G1,m.G2)]))
run('inf,G2', pr([(None,m.G2)]))
run('G1,inf', pr([(m.G1,None)]))
run('inf,inf', pr([(None,None)]))
run('2 pairs->1', pr([(m.g1_mul(m.G1,3),m.G2),(m.ec_neg(m.FP,m.G1),m.g2_mul(m.G2,3))]))
P5=m.g1_mul(m.G1,5); Q7=m.g2_mul(m.G2,7)
run('5G1,7G2', pr([(P5,Q7)]))
