"""CPU ORACLE — test infrastructure only.

A CPU restatement (numpy / torch-CPU / plain C) of the reference's algorithm for the quantized CenterPoint
hot path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and only as
the checker. The product package never imports anything from here.

Pinning status: the reference ships no tests, golden vectors or fixtures for this path, and
its third-party arithmetic (spconv, pytorch_quantization) is absent from the container. The dense SmoothQuant
conv and the BEV backbone restatements are pinned against the reference's own classes imported in the build
container (tests/golden/make_golden.py → tests/golden/*.npz); the voxelizer, sparse-conv rulebook and
TensorQuantizer restatements follow the library semantics listed in SURVEY.md §8c and are **parity unpinned**.
"""
