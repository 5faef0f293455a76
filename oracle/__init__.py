"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Restates the reference's algorithms for the hot path so the HIP path can be
checked on a box where /root/reference does not exist.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package.  The product (artspeech_amd/) never does.
Pinning: every function here is checked against outputs of the reference itself, generated in the
build container by tests/golden/make_golden.py and committed under tests/golden/.
"""
