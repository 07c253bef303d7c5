"""the cpu_baseline leg of bench.py on its own (host cores only)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "NS"
C = {"NS": 256, "S": 32, "St": 32}[name]
print(json.dumps(bench.cpu_baseline(name, 500000, C)))
