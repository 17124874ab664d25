e	a	1	1	1
e	b	1	1	1
a	c	1	1	2
a	d	1	1	2
h	f	1	2	1
f	g	1	2	2
