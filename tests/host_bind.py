"""The ctypes binding of libfamsa_host.so lives in the package (famsa_amd/hostlib.py); the tests use it from here."""
from famsa_amd.hostlib import CLI, DIST, HOST_SO, Host  # noqa: F401
