"""Drop-in `phe` package backed by hand-written HIP kernels for AMD MI355X (gfx950).

Same public names as the reference package (phe/__init__.py:1-8 of data61/python-paillier 1.5.0).
"""
__version__ = "1.5.0+mi355x.r1"
