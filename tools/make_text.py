#!/usr/bin/env python3
"""Write a text-like file (tests/corpus.text_like) of the given size: python tools/make_text.py <bytes> <path>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import corpus
corpus.text_like(int(sys.argv[1]), 0x5EED0004).tofile(sys.argv[2])
