"""BASELINE config 4: Whisper-small encoder + CTC head train step on one MI355X (tools/side_benches.py: whisper_step; bench.py
reports the same line under `secondary`).  usage: python tools/bench_whisper.py [B=8] [steps=10]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from side_benches import whisper_step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
print(json.dumps(whisper_step(B, steps)))
