"""efts_resconv5_multi: time of a grouped launch (mel-length layer + text-length layer) against single launches (us)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_resconv_gpu import Case, C, _dev
from efficient_tts_amd import ops as P

def timeit(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def kw_of(c, split):
    dev = _dev()
    y = P.Plane.for_rows(c.rs, C, split, dev)
    yl = P.Plane.for_rows(c.rs, C, 1, dev) if split == 1 else None
    return dict(x=c.a, x_lo=c.a_lo, w=c.pw, m=c.rs.rows, n=C, bias=c.bias, rowmask_ptr=c.mask.data_ptr(), y=y, y_lo=yl)

for split in (1, 2):
    mel, txt, big = Case(64, 800, split), Case(64, 128, split, seed=5), Case(64, 930, split, seed=9)
    km, kt, kb = kw_of(mel, split), kw_of(txt, split), kw_of(big, split)
    print(f"split {split}")
    for rep in range(2):
        print(f"  mel alone (51328 rows)        : {timeit(lambda: P.resconv5(**km)):7.1f} us   plan {P.resconv5_plan(mel.rs.rows, C)}")
        print(f"  text alone (8320 rows)        : {timeit(lambda: P.resconv5(**kt)):7.1f} us   plan {P.resconv5_plan(txt.rs.rows, C)}")
        print(f"  one layer of 59648 rows       : {timeit(lambda: P.resconv5(**kb)):7.1f} us   plan {P.resconv5_plan(big.rs.rows, C)}")
        print(f"  mel + text grouped            : {timeit(lambda: P.resconv5_multi([km, kt])):7.1f} us")
        for classes in ([[8, 7], [7, 8]], [[8, 7]], [[7, 8]], [[5, 5, 5]], [[6, 6, 4], [4, 6, 6]], [[8, 8], [8, 7]], [[6, 6, 3], [6, 3, 6]]):
            try:
                k0 = dict(km); k0["plan"] = P.make_plan(mel.rs.rows + txt.rs.rows, classes)
                print(f"  grouped, plan {str(classes):28s}: {timeit(lambda: P.resconv5_multi([k0, kt])):7.1f} us  groups {k0['plan'][0]}")
            except Exception as ex:
                print(f"  grouped, plan {classes}: {ex}")
