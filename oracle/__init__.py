"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package.  ``cosa_amd`` never does (tests/test_layout.py enforces it).

Contents
  cosa_oracle.c      plain-C restatement of the label / PAR / CAM-norm / bilateral stages
  c_oracle.py        ctypes binding to liboracle.so (+ to oracle/_ref/libref_bilateral.so,
                     the reference's own C++ compiled from /root/reference, when present)
  torch_oracle.py    torch-CPU fp32 restatement of the network + losses (floating point)
  ref_loader.py      loads the REFERENCE's Python files by path (authoring container only)
  gen_golden.py      writes tests/golden/*.npz from the reference (authoring container only)

Parity pinning: the reference has no tests; the oracle is pinned by golden vectors
produced by the reference itself (gen_golden.py), committed under tests/golden/.
"""
