"""CPU oracle for the RoITr hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (roitr_amd/) never does; it fails loudly without its HIP library.
"""
