"""`python -m moge_amd.scripts.cli <command>` - the reference's `moge` command group (moge/scripts/cli.py:10-24, pyproject.toml:36) with the
commands that exist on this path: `infer`, `infer_panorama`, `infer_baseline`.  `app` (Gradio), `eval_baseline` (the metric harness - its
plugin side is baselines/moge_mi355x.py, its alignment solvers moge_amd.alignment), `train` and `vis_data` are outside the hot path
 and are run from the reference checkout with the plugin."""
import click


@click.group(help="MoGe command line interface (MI355X path).")
def cli():
    pass


def main():
    from moge_amd.scripts import infer, infer_baseline, infer_panorama
    cli.add_command(infer.main, name="infer")
    cli.add_command(infer_baseline.main, name="infer_baseline")
    cli.add_command(infer_panorama.main, name="infer_panorama")
    cli()


if __name__ == "__main__":
    main()
