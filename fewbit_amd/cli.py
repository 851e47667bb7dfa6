"""``python -m fewbit quantize <bits> <module:func>``: build a few-bit table for an activation's derivative and
store it in the npz format ``StepwiseStore.load`` reads (reference: fewbit/cli.py, tools/quantize-builtins.sh)."""
import logging
from argparse import ArgumentParser
from importlib import import_module
from pathlib import Path
from sys import stderr
from typing import Optional

from .approx import approximate

__all__ = ('main', 'quantize')


def quantize(nobits: int, spec: str, output: Optional[Path] = None, seed: Optional[int] = None, max_iters: int = 10000,
             border_error: float = 1e-6, level_error: float = 1e-6):
    """Fit ``2**nobits`` levels to the derivative of ``module:func`` (a torch function); returns the StepWiseFunction."""
    import numpy as np
    import torch

    module_name, func_name = spec.split(':', 1)
    func = getattr(import_module(module_name), func_name)

    def primitive(xs: np.ndarray) -> np.ndarray:
        return func(torch.tensor(xs)).numpy()

    def derivative(xs: np.ndarray) -> np.ndarray:
        ps = torch.tensor(xs, requires_grad=True)
        func(ps).backward(torch.ones_like(ps))
        return ps.grad.numpy()

    logging.info('quantizing the gradient of %s with %d bits', spec, nobits)
    quant, info = approximate(fn=derivative, fn_prim=primitive, cardinality=2**nobits, parity=False, max_iters=max_iters,
                              beps=border_error, leps=level_error, domain=(-100, 100), random_state=seed)
    if info['status'] != 'converged':
        logging.error('failed to converge in %d iterations (%s)', info['noiters'], info['status'])
        raise SystemExit(1)
    logging.info('converged in %d iterations\n%s', info['noiters'], quant)

    if output is not None:
        case = f'{func_name}{nobits:02d}'
        arrays = {}
        if Path(output).exists():
            try:
                with np.load(output) as npz:
                    arrays = dict(npz)
            except Exception:  # noqa: BLE001
                logging.error('failed to load existing file %s: overwrite it', output)
        arrays[f'{case}-borders'] = quant.borders
        arrays[f'{case}-levels'] = quant.levels
        with open(output, 'wb') as fout:                    # np.savez would append .npz to a bare name
            np.savez(fout, **arrays)
        logging.info('saved to %s', output)
    return quant


def build_parser() -> ArgumentParser:
    parser = ArgumentParser(prog='fewbit', description=__doc__)
    parser.set_defaults(cmd=None)
    parser.add_argument('--log-level', default='info', choices=('debug', 'error', 'info', 'warn'))
    sub = parser.add_subparsers()
    sub.add_parser('help', add_help=False, help='Show this message and exit.').set_defaults(cmd='help')
    sub.add_parser('version', add_help=False, help='Show version information.').set_defaults(cmd='version')
    q = sub.add_parser('quantize', help='Build and save few-bit approximation.')
    q.set_defaults(cmd='quantize')
    q.add_argument('-M', '--max-iters', type=int, default=10000)
    q.add_argument('-b', '--border-error', type=float, default=1e-6)
    q.add_argument('-l', '--level-error', type=float, default=1e-6)
    q.add_argument('-o', '--output', type=Path, default=None, help='npz file to create or update')
    q.add_argument('-s', '--seed', type=int, default=None)
    q.add_argument('nobits', type=int, help='Number of bits to use in quantization.')
    q.add_argument('spec', type=str, help='Qualified name of function to quantize (e.g. "torch.nn.functional:gelu").')
    return parser


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    logging.basicConfig(format='%(asctime)s %(levelname)s %(message)s', stream=stderr,
                        level={'debug': logging.DEBUG, 'info': logging.INFO, 'warn': logging.WARN,
                               'error': logging.ERROR}[args.log_level])
    if args.cmd is None:
        parser.print_usage()
    elif args.cmd == 'help':
        parser.print_help()
    elif args.cmd == 'version':
        from . import __version__
        print(f'fewbit version {__version__}')
    else:
        quantize(args.nobits, args.spec, args.output, args.seed, args.max_iters, args.border_error, args.level_error)
