"""Micro-benchmark of the BatchNorm backward forms on the MI355X: reduce + apply (three launches) against the persistent
one-pass kernel (ledn_bn_act_bwd_fused), back-to-back calls timed with HIP events, cold (a 600 MB fill between calls)
and warm.   python tools/bn_fused_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch  # noqa: E402

from test_stream_fast import _bn_bwd_case  # noqa: E402


def main():
    from led_net_amd import ops, ops_train as T
    dev = torch.device('cuda:0')
    flush = torch.empty(150 << 20, dtype=torch.float32, device=dev)
    for (C, P, act, res_mode, dres) in [(64, 262144, ops.ACT_PRELU, ops.RES_NONE, False), (64, 262144, ops.ACT_PRELU, ops.RES_ADD, True),
                                        (128, 65536, ops.ACT_PRELU, ops.RES_NONE, False), (16, 262144, ops.ACT_PRELU, ops.RES_NONE, False),
                                        (32, 262144, ops.ACT_RELU, ops.RES_NONE, False), (64, 65536, ops.ACT_PRELU, ops.RES_ADD, True),
                                        (128, 16384, ops.ACT_PRELU, ops.RES_NONE, False)]:
        kw = _bn_bwd_case(dev, C, P, act, res_mode, dres, False, 1)
        z, dy = kw.pop('z'), kw.pop('dy')
        row = []
        for fused in (0, 1):
            T.set_bn_fused(fused)
            for cold in (False, True):
                ts = []
                for it in range(12):
                    sinks = (torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev))
                    if cold:
                        flush.fill_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    T.bn_act_bwd(z, dy, sinks=sinks, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                ts = sorted(ts[2:])
                row.append(ts[len(ts) // 2])
        T.set_bn_fused(0)
        print(f'C{C:4d} P{P:7d} act{act} res{res_mode}: two-kernel warm {row[0]:6.1f} us cold {row[1]:6.1f} us | fused warm {row[2]:6.1f} us cold {row[3]:6.1f} us')


if __name__ == '__main__':
    main()
