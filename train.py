#!/usr/bin/env python
"""Entry point kept at the reference's location (`./train.py -c 18 -g 0 ...`); the code lives in the package."""
from zeroshotsemanticsegmentation_amd.train import main

if __name__ == '__main__':
    main()
