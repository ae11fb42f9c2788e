#!/usr/bin/env python3
"""Diagnosis: one RCCL communicator whose collectives are captured on a FORKED stream of a hipGraph capture.
usage: python tools/diag_comm_stream.py <variant>   (a: plain fork; b: eager op on the comm stream first; c: two ops, second
stream hop from another forked stream; d: grouped; e: the hop target is the capture's origin stream)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
importlib.import_module('led_net_amd')
from led_net_amd import rccl

v = sys.argv[1] if len(sys.argv) > 1 else 'a'
dev = torch.device('cuda:0')
comm = rccl.Comm(0, 1, dev)
cs = torch.cuda.Stream(device=dev)
br = torch.cuda.Stream(device=dev)
x = torch.ones(1024, device=dev)
torch.cuda.synchronize()
MAIN = [torch.cuda.current_stream(dev)]


def hop(t):
    global cs
    cur = torch.cuda.current_stream(dev)
    if v == 'e':
        cs = MAIN[0]
    if cur == cs:
        comm.all_reduce(t, t)
        return
    cs.wait_stream(cur)
    with torch.cuda.stream(cs):
        if v == 'd':
            with comm.group():
                comm.all_reduce(t, t)
                comm.all_reduce(t, t)
        else:
            comm.all_reduce(t, t)
    cur.wait_stream(cs)


if v in ('b', 'c', 'd', 'e'):
    hop(x)
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    MAIN[0] = torch.cuda.current_stream(dev)
    y = x * 2
    hop(y)
    if v in ('c', 'e'):
        cur = torch.cuda.current_stream(dev)
        br.wait_stream(cur)
        with torch.cuda.stream(br):
            w = y + 3
            hop(w)
            w = w * 2
            hop(w)
        cur.wait_stream(br)
        y = y + w
    z = y + 1
print('captured', v, flush=True)
g.replay()
torch.cuda.synchronize()
print('replayed', v, float(z[0]), flush=True)
