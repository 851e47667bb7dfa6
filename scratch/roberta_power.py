"""Package power and shader clock (sysfs hwmon / pp_dpm_sclk, sampled every ~5 ms by a thread) while the arms of scratch/roberta_ab.py
run, 30 steps each: does the S-from-memory path make an fp32 step slower by pushing the package to its power limit?
   python scratch/roberta_power.py fp32|bf16"""
import glob, os, statistics, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import fewbit
from fewbit_amd import cabi
import roberta_bench as rb

which = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16}[which]
dev = torch.device('cuda:0')

# ---- what the box lets an ordinary user read
cards = sorted(glob.glob('/sys/class/drm/card*/device'))
print('# sysfs devices:', cards)
power_files, sclk_files = [], []
for c in cards:
    power_files += glob.glob(c + '/hwmon/hwmon*/power1_average') + glob.glob(c + '/hwmon/hwmon*/power1_input')
    sclk_files += glob.glob(c + '/pp_dpm_sclk')
caps = glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap')
print('# power files:', power_files, 'cap files:', caps, 'sclk files:', sclk_files)
for f in caps:
    try:
        print('# cap', f, int(open(f).read()) / 1e6, 'W')
    except Exception as e:  # noqa: BLE001
        print('# cap unreadable', e)


def read_power():
    best = None
    for f in power_files:
        try:
            v = int(open(f).read()) / 1e6
            best = v if best is None else max(best, v)
        except Exception:  # noqa: BLE001
            pass
    return best


def read_sclk():
    best = None
    for f in sclk_files:
        try:
            for line in open(f):
                if '*' in line:
                    v = int(line.split(':')[1].strip().split('M')[0])
                    best = v if best is None else max(best, v)
        except Exception:  # noqa: BLE001
            pass
    return best


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.p, self.c = False, [], []

    def run(self):
        while not self.stop:
            p, c = read_power(), read_sclk()
            if p is not None:
                self.p.append(p)
            if c is not None:
                self.c.append(c)
            time.sleep(0.005)


g = torch.Generator().manual_seed(1)
ids = torch.randint(5, 50000, (128, 128), generator=g).to(dev)
labels = torch.randint(0, 2, (128,), generator=g).to(dev)
vanilla = rb.build(dtype, dev)
rnd = rb.build(dtype, dev)
rb.swap_linear(rnd, 0.2, None, 'gaussian')
layers = [m for m in rnd.modules() if isinstance(m, fewbit.RandomizedLinear)]


def steps(model, n=30, warm=3):
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        model(input_ids=ids, labels=labels).loss.backward()
        opt.step()

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    s = Sampler(); s.start()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    s.stop = True; s.join()
    return dt, s


def arm(kind, mem, p16):
    for m in layers:
        m.matmul = kind
    cabi.tune_sketch_materialise(mem)
    cabi.tune_sketch_partials(p16)
    return steps(rnd)


arms = {'vanilla': lambda: steps(vanilla), 'gaussian, S from memory (forced)': lambda: arm('gaussian', 1, -1), 'gaussian, fused (forced)': lambda: arm('gaussian', 0, -1),
        'rademacher': lambda: arm('rademacher', -1, -1)}
print(f'# RoBERTa-base b128 x s128 {which}: ms per step over 30 steps; package power (W) and shader clock (MHz) sampled every ~5 ms during those steps: mean (min..max), samples')
for rnd_i in range(2):
    for k, f in arms.items():
        dt, s = f()
        pw = f'{statistics.mean(s.p):7.1f} ({min(s.p):6.1f}..{max(s.p):6.1f}) n={len(s.p)}' if s.p else 'n/a'
        ck = f'{statistics.mean(s.c):7.1f} ({min(s.c)}..{max(s.c)}) n={len(s.c)}' if s.c else 'n/a'
        print(f'round {rnd_i} {k:36s} {dt:7.2f} ms   power {pw}   sclk {ck}', flush=True)
