"""The legs of bench.py (the driver records the sha of bench.py, the entry file; these modules hold what it measures outside the timed region)."""
