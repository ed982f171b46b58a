"""Support code of bench.py (measurement only; nothing here is imported by the package or by the tests' product paths)."""
