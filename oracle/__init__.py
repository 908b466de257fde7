"""TEST INFRASTRUCTURE — CPU oracle for the DP3 hot path (see numpy_ref.py, pcp_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product path (point-cloud-processing_amd/) must never import it."""
