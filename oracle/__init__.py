"""CPU oracle for the U-Net hot path -- TEST INFRASTRUCTURE ONLY.

A functional restatement (stock torch CPU fp32 ops, autograd for gradients) of the
reference's algorithm for the path named by BASELINE.json.  Every function cites the
reference file:line it follows.  The oracle is pinned against golden vectors produced by
importing the real reference (tests/golden/make_goldens.py, tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  Nothing under unet_amd/ imports it: the product path is the HIP library and it
fails loudly when that library is missing.
"""
