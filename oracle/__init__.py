"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU oracles for the residual/smoother hot path:
  * oracle/_ref/libadflow_ref.so  — the reference's own Fortran kernels, built
    in place from /root/reference by oracle/refbuild/Makefile (git-ignored).
  * oracle/adflow_oracle.c        — plain-C restatement (each function cites the
    reference file:line it follows), pinned against the above.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (adflow_amd/) never does.
"""
