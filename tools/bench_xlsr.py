"""BASELINE config 5 on one GPU: Wav2Vec2-large-XLSR CTC train step on mixed-length, length-grouped batches
(tools/side_benches.py: xlsr_bucketed; bench.py reports the same line under `secondary`).
usage: python tools/bench_xlsr.py [B=16] [N=320]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from side_benches import xlsr_bucketed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 320
print(json.dumps(xlsr_bucketed(B, N)))
