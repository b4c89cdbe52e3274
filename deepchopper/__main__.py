"""`python -m deepchopper ...` (deepchopper/__main__.py in the reference)."""
from deepchopper_amd.cli import main

if __name__ == "__main__":
    raise SystemExit(main())
