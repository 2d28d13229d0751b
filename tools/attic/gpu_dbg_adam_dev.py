import sys, ctypes as C, torch
sys.path.insert(0, ".")
from efficient_tts_amd import lib as L, ops as O
dev = torch.device("cuda:0")
lib = L.load(); L.require_device()
n = 1 << 20
torch.manual_seed(0)
p0, g = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-2
m0, v0, vm0 = torch.randn(n, device=dev) * 1e-3, torch.rand(n, device=dev) * 1e-4, torch.rand(n, device=dev) * 1e-4
ss = torch.tensor([float((g * g).sum())], device=dev)
for step in (1, 2, 7, 1000):
    res = []
    for devmode in (False, True):
        p, m, v, vm = p0.clone(), m0.clone(), v0.clone(), vm0.clone()
        with O.stream_scope():
            if devmode:
                arr = (C.c_float * 3)()
                L.check(lib.efts_adam_hyper(3.3e-4, 0.9, 0.99, step, arr), "h")
                words = torch.zeros(8, dtype=torch.int32, device=dev)
                O.store_words(words, list((C.c_uint32 * 3).from_buffer(arr)) + [5])
                L.check(lib.efts_adam_amsgrad_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), vm.data_ptr(), n, ss.data_ptr(), 1.0, 1.0,
                                                  words.data_ptr(), 0.9, 0.99, 1e-9, 1e-5, O._stream()), "d")
            else:
                L.check(lib.efts_adam_amsgrad(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), vm.data_ptr(), n, ss.data_ptr(), 1.0, 1.0,
                                              3.3e-4, 0.9, 0.99, 1e-9, 1e-5, step, O._stream()), "e")
            torch.cuda.synchronize()
        res.append((p, m, v, vm))
    print(step, [bool(torch.equal(a, b)) for a, b in zip(*res)], float((res[0][0] - res[1][0]).abs().max()))
