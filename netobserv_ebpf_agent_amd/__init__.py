"""Importable alias of the package directory `netobserv-ebpf-agent_amd/` (a
hyphen is not a valid Python identifier). All code lives there."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "netobserv-ebpf-agent_amd"))
_here = __path__[0]
with open(_os.path.join(_here, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_here, "__init__.py"), "exec"))
del _f
