"""CPU oracle for the conditional-U-Net hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``weather-unet_amd/``) never does.
"""
