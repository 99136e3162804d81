"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU (plain PyTorch fp32) restatement of the CP-CSV story-GAN training step of
basiclab/CPCStoryVisualization-Pytorch, used as the parity checker for the HIP path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
the product package never does.
"""
