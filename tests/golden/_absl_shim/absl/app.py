"""absl.app stand-in."""
import sys


def run(main, argv=None):
    return main(sys.argv if argv is None else argv)
