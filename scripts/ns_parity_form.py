"""scratch: the full-NS parity test with the gather-once kernel forced to one form (0 = round 4, 1 = round 5): python scripts/ns_parity_form.py 0"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from cnrma_amd import sparse as S
S.conv_tuning(go=int(sys.argv[1]))
sys.exit(pytest.main([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_fullsize_oracle_gpu.py"),
                      "-q", "-s", "-k", "north_star_network"]))
