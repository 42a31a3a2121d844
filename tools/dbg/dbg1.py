import os, sys, subprocess
sys.path.insert(0, os.getcwd())
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_fullsize.py", "-q", "-x", "-k", "xlsr_large or adamw", "-s"], capture_output=True, text=True)
out = r.stdout
i = out.find("FAILURES")
print(out[i:i + 6000] if i >= 0 else out[-3000:])
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_fullsize.py", "-q", "-k", "adamw", "-s"], capture_output=True, text=True)
out = r.stdout
i = out.find("FAILURES")
print(out[i:i + 4000] if i >= 0 else out[-1500:])
