"""Clocks and socket power WHILE the training step runs: a sampler thread calls rocm-smi every ~0.4 s during a few thousand steps.
(Round 4: is the step running at the chip's power budget?  The kernel-level PMC passes of round 3 read an effective 1.78 GHz under
the head convolution alone.)  Usage: python tools/clock_probe.py [seconds]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dsl_amd import detectors  # noqa: F401
from dsl_amd.data import mark_ready
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
model = build_detector(bench.model_cfg()).cuda()
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
batch = bench.synth_batch(0, 2)
ev = torch.cuda.Event(); ev.record()
def step():
    mark_ready(batch['img'], event=ev); out = model.train_step(batch, opt); out['loss'].backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showuse', '--showtemp'], capture_output=True, text=True, timeout=10).stdout
            g = lambda pat: (re.search(pat, o) or [None, None])[1]
            samples.append(dict(t=time.perf_counter(), sclk=g(r'sclk clock level: \S+ \((\d+)Mhz\)'), mclk=g(r'mclk clock level: \S+ \((\d+)Mhz\)'),
                                fclk=g(r'fclk clock level: \S+ \((\d+)Mhz\)'), power=g(r'Power \(W\): ([\d.]+)'), use=g(r'GPU use \(%\): (\d+)'),
                                temp=g(r'Temperature \(Sensor junction\) \(C\): ([\d.]+)')))
        except Exception as e:  # noqa: BLE001
            samples.append(dict(err=str(e)))
        time.sleep(0.3)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20): step()
    n += 20
    if n % 200 == 0: torch.cuda.synchronize()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
stop = True; th.join()
print(f'{n} steps in {dt:.2f} s = {2 * n / dt:.1f} img/s')
for s_ in samples: print({k: v for k, v in s_.items() if k != 't'})
idle = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
print('idle after the run:', re.findall(r'sclk clock level: \S+ \((\d+)Mhz\)', idle), re.findall(r'Power \(W\): ([\d.]+)', idle))
