"""Stands where the reference imports the third-party `opty` package (src/single_opt_planner.py:12, src/multi_opt_planner.py:12,
src/06_optyplan.py, src/07_multioptyplan.py): only `opty.direct_collocation.Problem`, the one class the reference uses, solved on
the GPU by libd2dhip's collocation-NLP kernel.  Put this package's parent directory on sys.path ahead of a real opty to use it."""
from . import direct_collocation   # noqa: F401
