"""Which FLOAT atomic sites does a training step go through?  Runs two bf16 (or f32) steps of the product on the CPU
emulator build of the kernel sources (tests/emu/libledn_emu.so) with the emulator's census of float atomicAdd call
sites switched on, and prints "source:line count" per site.  With LEDN_DETERMINISTIC=1 the list must be empty
(tests/test_deterministic.py asserts it).   usage: python tools/atomic_census.py [bf16|f32] [HxW] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402


def census(dtype='bf16', hw=(320, 320), nb=1, steps=2, out='/tmp/ledn_atomic_census.txt'):
    import led_net_amd as L
    from conftest import bind_emu
    with bind_emu() as lib:
        lib.cdll.ledn_emu_atomic_census(None, 1)
        torch.manual_seed(304)
        cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
        for c in cfg['model']['decode_head']['loss_decode']:
            c['min_kept'] = 20000
        model = L.MODELS.build(cfg['model'])
        if dtype == 'bf16':
            model.set_act_dtype(torch.bfloat16)
        g = torch.Generator().manual_seed(5)
        img = torch.randint(0, 256, (nb, 3, *hw), dtype=torch.uint8, generator=g)
        lab = torch.randint(0, 2, (nb, 1, *hw), dtype=torch.int64, generator=g)
        lab[:, :, :5, :] = 255
        samples = [L.SegDataSample(gt=lab[i]) for i in range(nb)]
        tr = L.Trainer(model, cfg, max_iters=100)
        for _ in range(steps):
            tr.train_step(img, samples)
        n = lib.cdll.ledn_emu_atomic_census(out.encode(), 1)
    return n, open(out).read()


if __name__ == '__main__':
    dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    hw = tuple(int(v) for v in sys.argv[2].split('x')) if len(sys.argv) > 2 else (320, 320)
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    n, txt = census(dt, hw, nb)
    print(f'{n} float-atomic sites reached ({dt}, {hw[0]}x{hw[1]}, batch {nb}, LEDN_DETERMINISTIC={os.environ.get("LEDN_DETERMINISTIC", "0")}):')
    print(txt)
