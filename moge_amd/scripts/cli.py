"""`python -m moge_amd.scripts.cli <command>` - the reference's `moge` command group (moge/scripts/cli.py:10-24, pyproject.toml:36) with the
commands that exist on this path: `infer`, `infer_panorama`, `infer_baseline`.  `app` (Gradio), `eval_baseline` (the metric harness - its
plugin side is baselines/moge_mi355x.py, its alignment solvers moge_amd.alignment), `train` and `vis_data` are outside the hot path
 and are run from the reference checkout with the plugin."""
import importlib

import click

COMMANDS = ("infer", "infer_baseline", "infer_panorama")      # module moge_amd.scripts.<name>, click command `main`


class _LazyGroup(click.Group):
    """Sub-commands are imported when asked for: `--help` of the group does not pull in torch."""

    def list_commands(self, ctx):
        return sorted(COMMANDS)

    def get_command(self, ctx, name):
        if name not in COMMANDS:
            return None
        return importlib.import_module(f"moge_amd.scripts.{name}").main


cli = _LazyGroup(name="moge", help="MoGe command line interface (MI355X path).")


def main():
    cli()


if __name__ == "__main__":
    main()
