"""The pieces of bench.py (round 5: 1500 lines in one file became the contract line + these modules)."""
