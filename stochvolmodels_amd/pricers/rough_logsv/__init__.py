"""host-side set-up of the rough (Markovian lift) LogSV simulator: the kernel quadrature rule"""
