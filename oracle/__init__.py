"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline)."""
