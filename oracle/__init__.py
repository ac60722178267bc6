"""CPU oracle for the read-generation hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (insilicoseq_amd/) never does."""
