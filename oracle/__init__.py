"""TEST INFRASTRUCTURE ONLY -- the CPU oracle for the PLIP embedding hot path.

Nothing under ``plip_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / the timed CPU baseline -- never as the product path.
"""
