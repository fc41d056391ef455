"""Module `bsqpN32_iiwa14` of the reference's build (CMakeLists.txt:46-61): KNOT_POINTS = 32, classes `BSQP_{B}_float`."""
from ._module_factory import populate

populate(globals(), "iiwa14", 32)
