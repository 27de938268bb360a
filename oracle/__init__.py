"""ORACLE package: test infrastructure only (CPU restatements + reference import tooling).

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under tcow_amd/ imports it.
"""
