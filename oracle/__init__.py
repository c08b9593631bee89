"""CPU oracle for the MockingBird hot path -- TEST INFRASTRUCTURE ONLY.

A restatement of the reference's algorithm for the mel-synthesis + vocoder
forward path, function by function, each citing the reference file:line it
follows.  The reference computes in fp32 through PyTorch's ATen CPU kernels
(there is no other arithmetic in it), so the restatement uses
the same ATen calls (torch.nn.functional on CPU tensors) in the same order;
byte/index work (mu-law, fold/unfold, monotonic path) is numpy / plain C.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, generated in the
build container by tests/golden/make_golden.py (which imports
/root/reference) and committed under tests/golden/*.npz;
tests/test_oracle_golden.py checks every oracle function against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  Nothing under mockingbird_amd/ does.
"""
