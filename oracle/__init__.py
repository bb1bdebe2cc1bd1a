"""CPU oracle for the Assemble-ResNet training path (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product package ``assembled_cnn_amd`` never
does; it fails loudly when its HIP library is missing.
"""
