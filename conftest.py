# The reference's entry-point script keeps its name (test_nbp_planning.py); it is not a pytest module.
collect_ignore = ["test_nbp_planning.py", "train_nbp.py", "bench.py"]
