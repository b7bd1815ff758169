"""The legs of bench.py (the driver keeps the CLI, the order of the legs and the one JSON line)."""
