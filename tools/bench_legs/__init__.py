"""Side legs of bench.py, one module per object of the JSON line (bench.py keeps the headline and the assembly
of the line).  Every leg is `run(env, ...) -> dict | None`; `env` is the Env bench.py builds once (device, context, handles, rank
plumbing).  The oracle is used by the legs only as checker of sampled outputs and by cpu_baseline as the thing it times."""
