"""The C-ABI library loads and exports every symbol include/ledn.h declares
(no compute calls: works without a GPU), and the product path refuses to run
without it / on CPU tensors."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'ledn.h')).read()
    return sorted(set(re.findall(r'^(?:int|long long) (ledn_\w+)\(', src, flags=re.M)))


def test_header_and_binding_agree():
    from led_net_amd import _lib
    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_hip_library_exports_every_symbol():
    import __graft_entry__ as g
    from led_net_amd import _lib
    g.build(full=False)          # incremental make: hipcc cross-compiles gfx950 without a GPU
    lib = _lib.Library(_lib.HIP_LIB_PATH, is_hip=True)
    for name in declared_symbols():
        assert hasattr(lib.cdll, name), name
    assert lib.cdll.ledn_abi_version() == _lib.ABI_VERSION == 5
    src = open(os.path.join(ROOT, 'include', 'ledn.h')).read()
    assert f'#define LEDN_ABI_VERSION {_lib.ABI_VERSION}' in src


def test_product_path_has_no_cpu_fallback():
    """CPU tensors handed to the product ops must raise, not silently compute."""
    import __graft_entry__ as g
    from led_net_amd import ops
    g.build(full=False)
    with pytest.raises(ops.LednError):
        ops.affine_act(torch.zeros(1, 2, 2, 4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'led-net_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dirpath, f)
                assert 'oracle.' not in src and 'oracle/' not in src, os.path.join(dirpath, f)


def test_reference_config_parses_unchanged():
    """configs/LED_Net/LEDNet_80k_cityscapes-1024x1024.py (model section mirrored in
    tests/data) builds through the registry with the reference's type names."""
    import led_net_amd as L
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    assert cfg['model']['backbone']['type'] == 'LEDNet'
    assert cfg['model']['decode_head']['type'] == 'LEDHead'
    m = L.MODELS.build(cfg['model'])
    assert isinstance(m.backbone, L.LEDNet) and isinstance(m.decode_head, L.LEDHead)
    assert [l.loss_weight for l in m.decode_head.loss_decode] == [1.0, 0.4]
    keys = set(m.state_dict())
    for k in ('decode_head.head.0.bn.weight', 'decode_head.head.0.conv.weight', 'decode_head.head.1.weight',
              'decode_head.conv_seg.bias', 'decode_head.aux_cls_seg.weight',
              'backbone.aff1.local_att.0.weight', 'backbone.layer5_.spp_dw.0.conv.weight'):
        assert k in keys, k
    ref = '/root/reference/configs/LED_Net/LEDNet_80k_cityscapes-1024x1024.py'
    if os.path.exists(ref):          # this container only
        assert L.load_config(ref)['model'] == cfg['model']


def test_register_into_mmseg_with_a_stub_registry(monkeypatch):
    """register_into_mmseg() against a stand-in for mmseg.registry.MODELS with mmengine's register_module signature
    (name=, force=, module=): the three plugin classes land under the reference's type names
    (backbones/__init__.py:29,37, decode_heads/led_head.py:15, losses/ohem_cross_entropy_loss.py) and build from a
    config dict the way MODELS.build(cfg.model) would hand them their arguments (encoder_decoder.py:89,102)."""
    import sys
    import types
    import led_net_amd as L
    from led_net_amd import registry

    class StubRegistry:
        def __init__(self):
            self.module_dict = {}

        def register_module(self, name=None, force=False, module=None):
            assert module is not None and isinstance(module, type)
            if name in self.module_dict and not force:
                raise KeyError(name)
            self.module_dict[name] = module
            return module

        def build(self, cfg):
            cfg = dict(cfg)
            return self.module_dict[cfg.pop('type')](**cfg)

    assert registry.register_into_mmseg() is False or 'mmseg' in sys.modules        # not installed here: a no-op
    stub = StubRegistry()
    mm, reg = types.ModuleType('mmseg'), types.ModuleType('mmseg.registry')
    reg.MODELS = stub
    mm.registry = reg
    monkeypatch.setitem(sys.modules, 'mmseg', mm)
    monkeypatch.setitem(sys.modules, 'mmseg.registry', reg)
    assert registry.register_into_mmseg() is True
    assert set(stub.module_dict) == {'LEDNet', 'LEDHead', 'OhemCrossEntropy'}
    assert stub.module_dict['LEDNet'] is L.LEDNet and stub.module_dict['LEDHead'] is L.LEDHead
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))['model']
    bb = stub.build(cfg['backbone'])
    assert isinstance(bb, L.LEDNet) and bb.channels == 32
    loss = stub.build(cfg['decode_head']['loss_decode'][0])
    assert loss.loss_name == 'loss_context' or hasattr(loss, 'loss_name')
    assert registry.register_into_mmseg(force=True) is True                          # idempotent under force
    with pytest.raises(KeyError):
        registry.register_into_mmseg(force=False)


def test_two_streams_keep_their_own_workspace():
    """boundary state is keyed by stream (ledn_bind_workspace), not process-wide: binding a scratch buffer to one
    stream does not change what another stream's launches use (checked on the emulator build: no GPU needed)"""
    import ctypes as C
    from conftest import bind_emu
    from led_net_amd import ops
    with bind_emu() as lib:
        a = torch.zeros(1 << 16)
        b = torch.zeros(1 << 16)
        sa, sb = C.c_void_p(0x10), C.c_void_p(0x20)            # stream handles are opaque keys for the emulator
        assert lib.cdll.ledn_bind_workspace(sa, a.data_ptr(), a.numel()) == 0
        assert lib.cdll.ledn_bind_workspace(sb, b.data_ptr(), b.numel()) == 0
        x = torch.randn(2, 40, 40, 8).bfloat16()
        # a statistics producer with > 16 workgroups writes its partial rows into the CALLING stream's workspace
        st = (torch.zeros(8), torch.zeros(8))
        lib.call('ledn_channel_stats', x.data_ptr(), None, x.numel() // 8, 8, 1, st[0].data_ptr(), st[1].data_ptr(), sa)
        used_a, used_b = bool(a.abs().sum() > 0), bool(b.abs().sum() > 0)
        assert not used_b, 'a launch on stream A touched the workspace bound to stream B'
        torch.testing.assert_close(st[0], x.float().sum((0, 1, 2)), rtol=1e-3, atol=1e-2)
        assert lib.cdll.ledn_bind_workspace(sa, None, 0) == 0 and lib.cdll.ledn_bind_workspace(sb, None, 0) == 0
        del used_a


def test_set_workspace_alone_serves_unbound_streams():
    """ledn_set_workspace keeps the PROCESS DEFAULT scratch: a caller that never binds a stream (the C micro-benchmarks
    under tools/micro) must get its partial rows there -- round 3's setter only wrote the calling thread's snapshot,
    which the next entry point overwrote with the still-empty default."""
    import ctypes as C
    from conftest import bind_emu
    with bind_emu() as lib:
        ws = torch.zeros(1 << 16)
        s = C.c_void_p(0x30)                                     # a stream nobody bound
        assert lib.cdll.ledn_bind_workspace(s, None, 0) == 0
        assert lib.cdll.ledn_set_workspace(C.c_void_p(ws.data_ptr()), C.c_longlong(ws.numel())) == 0
        try:
            x = torch.randn(4, 64, 64, 8).bfloat16()         # 16 K pixels: the streaming statistics kernel, partial rows
            st = (torch.zeros(8), torch.zeros(8))
            lib.call('ledn_channel_stats', x.data_ptr(), None, x.numel() // 8, 8, 1, st[0].data_ptr(), st[1].data_ptr(), s)
            assert bool(ws.abs().sum() > 0), 'the default workspace was not used by a launch on an unbound stream'
            torch.testing.assert_close(st[0], x.float().sum((0, 1, 2)), rtol=1e-3, atol=5e-2)
            # the stem weight gradient writes its partial tiles unconditionally: without any workspace it must refuse
            assert lib.cdll.ledn_set_workspace(None, C.c_longlong(0)) == 0
            img = torch.zeros(1, 3, 16, 16, dtype=torch.uint8)
            dz = torch.zeros(1, 8, 8, 32).bfloat16()
            dw = torch.zeros(32, 3, 3, 3)
            rc = lib.cdll.ledn_stem_conv_wgrad(C.c_void_p(img.data_ptr()), 2, C.c_void_p(dz.data_ptr()),
                                               C.c_void_p(dw.data_ptr()), 1, 16, 16, 3, 8, 8, 32, None, None, None, None,
                                               C.c_float(0.0), s)
            assert rc == 1, 'ledn_stem_conv_wgrad without a workspace must return LEDN_EINVAL'
        finally:
            lib.cdll.ledn_set_workspace(None, C.c_longlong(0))
