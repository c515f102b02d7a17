"""CPU oracle for the registration hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under `oracle/` is product code.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import,
call, link or execute anything in here, and only as the *checker*.

PARITY UNPINNED: the reference (EdwardLiuyc/StaticMapping) ships no test,
golden vector or known-answer fixture for any registrator
and cannot be compiled in this image (Eigen / PCL / libnabo / libpointmatcher /
glog are absent), so these restatements are anchored on the reference's source
text (file:line cited per function) and on self-consistency KATs only.
`filters.py` (the pre-filters of SURVEY.md §8(f) N3) is the exception: the reference tests those, and
tests/test_oracle_filters.py pins the restatement on their known answers.
`ndt_gicp.py` (registrators::NdtWithGicp) additionally restates un-vendored PCL 1.8.1
(ApproximateVoxelGrid, NDT, GICP, BFGS) from its published algorithm; its header says which parts.
"""
