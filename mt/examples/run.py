"""`python -m mt.examples.run ...`: the reference's training entry point (mt/examples/run.py:28-186) on the HIP path."""
from mvae_amd.run import main

if __name__ == "__main__":
    main()
