"""CPU oracle — test infrastructure only (see oracle/icrec_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package; instacart_next_order_recommendation_amd never does.
"""
