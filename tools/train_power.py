"""GPU box: rocm-smi (socket power, shader clock) sampled while the training-step benchmark loops.  python tools/train_power.py [steps]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
samples, stop = [], False


def sampler():
    while not stop:
        try:
            samples.append((time.time(), subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout.strip().split("\n")[-1]))
        except Exception as e:      # noqa
            samples.append((time.time(), "ERR %r" % (e,)))
        time.sleep(0.1)


th = threading.Thread(target=sampler)
th.start()
steps = sys.argv[1] if len(sys.argv) > 1 else "60"
t0 = time.time()
out = subprocess.run([sys.executable, "bench.py", "--train", "--steps", steps, "--warmup", "3"], capture_output=True, text=True).stdout
t1 = time.time()
stop = True
th.join()
print(out.strip()[:400])
late = [s for t, s in samples if t > t0 + 0.6 * (t1 - t0)]
print("samples in the last 40 %% of the run: %d" % len(late))
for s in late[:: max(1, len(late) // 6)]:
    print(s)
