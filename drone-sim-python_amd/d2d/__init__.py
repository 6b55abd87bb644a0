"""Host-side mirror of the reference's `d2d` package for the planner / simulation hot path
(same module, class and method names).  Every numeric method that the
reference runs per drone per step goes through libd2dhip.so (d2dhip); nothing here falls
back to a CPU implementation when the library or the GPU is missing."""
