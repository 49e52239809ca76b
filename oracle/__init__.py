"""CPU oracle for the OptiState KF+GRU hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (optistate_amd) never imports it.  Parity status: pinned
against golden vectors generated from the reference (see tests/golden/ and tools/gen_golden.py).
"""
