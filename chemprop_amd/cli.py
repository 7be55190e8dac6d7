"""``chemprop-amd``: the reference's command line on the MI355X engine — ``chemprop_amd.enable()`` (rebinds the message-passing blocks
and ``MPNN`` the CLI names at import, ``cli/train.py:60-68``), then ``chemprop.cli.main.main()`` untouched.

    chemprop-amd train --data-path ... --devices 8          # == CHEMPROP_MI355X=1 chemprop train ... with the stub of INTEGRATION.md
    python -m chemprop_amd.cli predict ...
"""
from __future__ import annotations


def main() -> None:
    from .integration import enable

    enable()
    from chemprop.cli.main import main as chemprop_main  # noqa: WPS433  (the reference's entry point, as it is)

    chemprop_main()


if __name__ == "__main__":
    main()
