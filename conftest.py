"""pytest from the repository root: only tests/ holds tests.  experiments/ keeps measured-and-not-adopted kernels with the tests they were
checked with (they need code that is not compiled into the library); gpurun_out/ is scratch."""
collect_ignore_glob = ["experiments/*", "gpurun_out/*", "scripts/*", "profiles/*"]
